#!/bin/bash
# GPU call 2 of round 2: correctness of the rewritten scheduling + A/B of the new tunables
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > gpurun_out/r2c2_tests.log 2>&1
export GPK_X=$root/gpflow_amd/libgpk_exp.so
(GPK_LIBRARY=$GPK_X GPK_STREAM_PROJ=1 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fullsize.py -m gpu -q -x -k "svgp or column_groups" 2>&1 | tail -8) > gpurun_out/r2c2_tests_stream.log 2>&1
(GPK_LIBRARY=$GPK_X GPK_SOFT_RESERVE=0 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "svgp or column_groups" 2>&1 | tail -5) > gpurun_out/r2c2_tests_noresv.log 2>&1
tools/ab.sh "GPK_SOFT_RESERVE=0" "GPK_STREAM_PROJ=1" "GPK_STREAM_PROJ=1 GPK_SOFT_RESERVED_CUS=16" "GPK_STREAM_PROJ=1 GPK_SOFT_RESERVED_CUS=48" "GPK_SOFT_RESERVED_CUS=16" "GPK_SOFT_RESERVED_CUS=48" "GPK_STREAM_PROJ=1 GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=4" > gpurun_out/r2c2_ab.log 2>&1
python tools/gpr_predict_probe.py > gpurun_out/r2c2_pred.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $root/gpurun_out/r2c2_prof -o svgp -- python $root/tools/prof_run.py svgp > $root/gpurun_out/r2c2_prof.log 2>&1
GPK_LIBRARY=$GPK_X GPK_STREAM_PROJ=1 rocprofv3 --kernel-trace -d $root/gpurun_out/r2c2_prof_s -o svgp -- python $root/tools/prof_run.py svgp > $root/gpurun_out/r2c2_prof_s.log 2>&1
cd $root
for d in r2c2_prof r2c2_prof_s; do
  db=$(find gpurun_out/$d -name "*.db" | head -1)
  python tools/timeline.py $db rbf_kernel 4 140 > gpurun_out/${d}_timeline.txt 2>&1
  rm -rf gpurun_out/$d
done
cat gpurun_out/r2c2_tests.log | tail -6; cat gpurun_out/r2c2_ab.log; cat gpurun_out/r2c2_pred.log | tail -3
