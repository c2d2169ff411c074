#!/bin/bash
# Same-box A/B runs of bench.py under the tunables of the EXPERIMENTAL library (make -C gpflow_amd/csrc exp):
#   tools/ab.sh "GPK_STREAM_PROJ=1" "GPK_STREAM_PROJ=1 GPK_SOFT_RESERVED_CUS=16" ...
# The first line is always the product library with no tunables.
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  for rep in 1 2; do
    env $1 python $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpr --no-train --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('cfg=[$1] rep=$rep steps/s=%.1f ms=%.3f dom_us=%.0f big_gemm_TF=%.1f all_gemm_us=%.0f elbo=%.6f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['big_gemm_launches']['tflops_over_summed_durations'], r['all_gemm_launches']['avg_launch_us']*r['all_gemm_launches']['launches_per_step'], d['last_elbo']))" || echo "cfg=[$1] rep=$rep FAILED"
  done
}
run ""
for cfg in "$@"; do run "GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so $cfg"; done
