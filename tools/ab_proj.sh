#!/bin/bash
# A/B of the streamed q_sqrt projection on the SVGP step (same box):  tools/ab_proj.sh "ENV=.. ENV2=.." ...
run() {
  for rep in 1 2; do
    env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('cfg=[$1] rep=$rep steps/s=%.1f ms=%.3f big_gemm_TF=%.1f all_gemm_us=%.0f' % (d['value'], d['ms_per_step'], r['big_gemm_launches']['tflops_over_summed_durations'], r['all_gemm_launches']['avg_launch_us']*r['all_gemm_launches']['launches_per_step']))" || echo "cfg=[$1] rep=$rep FAILED"
  done
}
run "GPK_STREAM_PROJ=0"
for cfg in "$@"; do run "$cfg"; done
