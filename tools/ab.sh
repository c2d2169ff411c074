#!/bin/bash
# Same-box A/B runs of bench.py under the tunables of the EXPERIMENTAL library (make -C gpflow_amd/csrc exp):
#   tools/ab.sh "GPK_FLOW=0" "GPK_SOFT_RESERVED_CUS=48" ...
# The first line is always the product library with no tunables.  Every run is under a 75 s timeout.
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  for rep in $(seq 1 ${AB_REPS:-2}); do
    env $1 timeout 75 python $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpr --no-train --no-extras --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('cfg=[$1] rep=$rep steps/s=%.1f ms=%.3f dom=%s %.0fus allgemm_us=%.0f elbo=%.6f' % (d['value'], d['ms_per_step'], r['kernel'][:22], r['avg_launch_us'], r['all_gemm_launches']['avg_launch_us']*r['all_gemm_launches']['launches_per_step'], d['last_elbo']))" || echo "cfg=[$1] rep=$rep FAILED or TIMED OUT"
  done
}
run ""
for cfg in "$@"; do run "GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so $cfg"; done
