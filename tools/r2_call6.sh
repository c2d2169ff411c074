#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
export GPK_X=$root/gpflow_amd/libgpk_exp.so
(cd tools/r1_ref && for rep in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpr --no-train 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ROUND1 code rep=$rep steps/s=%.1f ms=%.3f' % (d['value'], d['ms_per_step']))"; done) > gpurun_out/r2c6_ab.log 2>&1
OLD="GPK_SOFT_RESERVE=0 GPK_GROUP_INVERSE=0 GPK_EXTRA_MAX_WGS=320"
tools/ab.sh "$OLD" "$OLD GPK_STREAM_LAYOUT=1" "$OLD GPK_STREAM_LAYOUT=2" "$OLD GPK_STREAM_LAYOUT=3" "$OLD GPK_STREAM_LAYOUT=2 GPU_MAX_HW_QUEUES=4" "GPK_STREAM_LAYOUT=2" "GPK_STREAM_LAYOUT=1" "GPK_STREAM_LAYOUT=3" "GPK_STREAM_LAYOUT=2 GPU_MAX_HW_QUEUES=4" >> gpurun_out/r2c6_ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
(cd $root/tools/r1_ref && rocprofv3 --kernel-trace -d $root/gpurun_out/r2c6_prof_r1 -o svgp -- python $root/tools/prof_run_r1.py svgp > $root/gpurun_out/r2c6_prof_r1.log 2>&1)
cd $root
for d in r2c6_prof_r1; do
  db=$(find gpurun_out/$d -name "*.db" | head -1)
  python tools/timeline.py $db rbf_kernel 4 120 > gpurun_out/${d}_timeline.txt 2>&1
  rm -rf gpurun_out/$d
done
cat gpurun_out/r2c6_ab.log
