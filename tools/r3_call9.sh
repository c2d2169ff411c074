#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
X="GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so"
( timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -5 ) > $out/r3c9_pytest.log 2>&1
export AB_REPS=2
bash tools/ab.sh "GPK_GROUP_FUSED=0" "GPK_GROUP_FUSED=1" "GPK_GROUP_FUSED=1 GPK_EXTRA_MAX_WGS=384" "GPK_GROUP_FUSED=0 GPK_EXTRA_MAX_WGS=384" "GPK_GROUP_FUSED=1 GPK_EXTRA_MAX_WGS=448" "GPK_GROUP_FUSED=1 GPK_XGROUP=384" > $out/r3c9_ab.log 2>&1
bash tools/ab_gpr.sh "GPK_GROUP_FUSED=0" "GPK_GROUP_FUSED=1" > $out/r3c9_ab_gpr.log 2>&1
bash tools/prof_timeline.sh r3c9_fused
