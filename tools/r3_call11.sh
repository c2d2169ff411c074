#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
X="GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so"
export AB_REPS=2
bash tools/ab.sh "GPK_XDEFER=1" "GPK_XDEFER=1 GPK_EXTRA_MAX_WGS=384" "GPK_XDEFER=1 GPK_EXTRA_MAX_WGS=448" "GPK_XDEFER=1 GPK_EXTRA_MAX_WGS=288" "GPK_XDEFER=0" > $out/r3c11_ab.log 2>&1
( env $X GPK_XDEFER=1 timeout 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_gradients.py -m gpu -x -q 2>&1 | tail -3 ) > $out/r3c11_pytest.log 2>&1
bash tools/prof_timeline.sh r3c11_defer $X GPK_XDEFER=1
