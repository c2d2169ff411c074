#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
export GPK_X=$root/gpflow_amd/libgpk_exp.so
(python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fullsize.py -m gpu -q -x -k "svgp or column_groups or potrf" 2>&1 | tail -4) > gpurun_out/r2c5_tests.log 2>&1
tools/ab.sh "GPK_SOFT_RESERVE=0 GPK_GROUP_INVERSE=0 GPK_EXTRA_MAX_WGS=320" "GPK_SOFT_RESERVE=0" "GPK_GROUP_INVERSE=0" "GPK_STREAM_PROJ=1" "GPK_STREAM_PROJ=1 GPK_SOFT_RESERVED_CUS=16" "GPK_SOFT_RESERVED_CUS=16" "GPK_SOFT_RESERVED_CUS=48" "GPU_MAX_HW_QUEUES=4" > gpurun_out/r2c5_ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $root/gpurun_out/r2c5_prof -o svgp -- python $root/tools/prof_run.py svgp > $root/gpurun_out/r2c5_prof.log 2>&1
GPK_LIBRARY=$GPK_X GPK_STREAM_PROJ=1 rocprofv3 --kernel-trace -d $root/gpurun_out/r2c5_prof_s -o svgp -- python $root/tools/prof_run.py svgp > $root/gpurun_out/r2c5_prof_s.log 2>&1
cd $root
for d in r2c5_prof r2c5_prof_s; do
  db=$(find gpurun_out/$d -name "*.db" | head -1)
  python tools/timeline.py $db rbf_kernel 4 170 > gpurun_out/${d}_timeline.txt 2>&1
  rm -rf gpurun_out/$d
done
tail -n 3 gpurun_out/r2c5_tests.log; cat gpurun_out/r2c5_ab.log
