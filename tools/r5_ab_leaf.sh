#!/bin/bash
# round 5, call 1: correctness of the fraction-free leaf, its phase timers, same-box A/B against the recurrence leaf
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
run() {
  for rep in 1 2; do
    env $1 timeout 150 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpr --no-train --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['other_workloads']
print('cfg=[$1] rep=$rep cm_ms=%.3f c3=%.3f c4_1024=%.3f c5sh=%.3f c5sep=%.3f unwh=%.3f qdiag=%.3f elbo=%.9e' % (d['ms_per_step'], o['c3']['ms_per_step'], o['c4_shard_1024']['ms_per_step'], o['c5_shared']['ms_per_step'], o['c5_separate']['ms_per_step'], o.get('cm_unwhitened',{}).get('ms_per_step',0), o.get('cm_q_diag',{}).get('ms_per_step',0), d['last_elbo']))" || echo "cfg=[$1] rep=$rep FAILED"
  done
}
run "" ; run "GPK_LIBRARY=$root/gpflow_amd/libgpk_oldleaf.so"
