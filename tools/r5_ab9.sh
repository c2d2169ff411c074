#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
AB_REPS=2 AB_FLAGS="--no-train --no-extras --no-other" bash tools/r5_ab_all.sh "GPK_TAIL_SPLIT=1" "GPK_TAIL_SPLIT=0" "GPK_TAIL_SPLIT_PCT=70" "GPK_TAIL_SPLIT_PCT=30"
