#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
X="GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so"
( timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 ) > $out/r3c13_pytest.log 2>&1
export AB_REPS=2
bash tools/ab.sh "GPK_GROUP_SOLVE_ROWS32=0" "GPK_GROUP_SOLVE_ROWS32=1" "GPK_GROUP_SOLVE_ROWS32=1 GPK_EXTRA_MAX_WGS=256" "GPK_GROUP_SOLVE_ROWS32=1 GPK_EXTRA_MAX_WGS=288" "GPK_GROUP_SOLVE_ROWS32=1 GPK_EXTRA_MAX_WGS=384" "GPK_GROUP_SOLVE_ROWS32=0 GPK_EXTRA_MAX_WGS=288" > $out/r3c13_ab.log 2>&1
bash tools/prof_timeline.sh r3c13_rows32
