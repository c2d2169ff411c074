#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
AB_REPS=3 AB_FLAGS="--no-train --no-extras --no-other" bash tools/r5_ab_all.sh "GPK_GPR_SPLIT_BUILD=1" "GPK_GPR_SPLIT_BUILD=0"
