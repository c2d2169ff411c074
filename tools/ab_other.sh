#!/bin/bash
# Same-box A/B of the OTHER BASELINE configs (C3, C4 shards, C5) under tunables of the experimental library:
#   tools/ab_other.sh "GPK_XGROUP=256" ...        (first line: product library, no tunables)
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  env $1 timeout 120 python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gpr --no-train --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('cfg=[$1] cm=%.3f ' % d['ms_per_step'] + ' '.join('%s=%.3f' % (k, v['ms_per_step']) for k, v in d['other_workloads'].items()))" || echo "cfg=[$1] FAILED"
}
run ""
for cfg in "$@"; do run "GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so $cfg"; done
