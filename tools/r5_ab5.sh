#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
timeout 300 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -2
AB_REPS=2 bash tools/r5_ab_all.sh "GPK_CHAIN_FLAGS_BATCHED=1" "GPK_CHAIN_FLAGS_BATCHED=0"
