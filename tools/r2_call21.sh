#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sgpr.py tests/test_gpu_gradients.py -m gpu -q -x 2>&1 | tail -4) > gpurun_out/r2c21_tests.log 2>&1
tail -n 2 gpurun_out/r2c21_tests.log
bash tools/ab_gpr.sh "GPK_NBO=768" "GPK_NBO=640 GPK_RESERVED_CUS=8" > gpurun_out/r2c21_gpr.log 2>&1; sed 's#GPK_LIBRARY=[^ ]* ##' gpurun_out/r2c21_gpr.log
F="GPK_FLOW=1 GPK_GROUP_INVERSE=1 GPK_SOFT_RESERVE=1 GPK_STREAM_PROJ=1"
tools/ab.sh "GPK_EXTRA_MAX_WGS=320" "$F GPK_FLOW_COH=2" "$F GPK_FLOW_COH=5" > gpurun_out/r2c21_ab.log 2>&1
grep "^cfg" gpurun_out/r2c21_ab.log | sed 's#GPK_LIBRARY=[^ ]* ##'
