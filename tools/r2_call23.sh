#!/bin/bash
# round-2 GPU call: tests after the identity-row / prologue changes, SVGP step A/B, training-step probe, GPR gradient
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
(time timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/r2c23_tests.log 2>&1
tail -n 5 gpurun_out/r2c23_tests.log
bash tools/ab.sh > gpurun_out/r2c23_ab.log 2>&1
cat gpurun_out/r2c23_ab.log
X=$root/gpflow_amd/libgpk_exp.so
(timeout 200 python tools/train_probe.py table gpr
 GPK_LIBRARY=$X GPK_HALF_TILE_BELOW=300 TRAIN_VARIANTS=default,no_overlap timeout 100 python tools/train_probe.py
 GPK_LIBRARY=$X GPK_HALF_TILE_BELOW=600 TRAIN_VARIANTS=default timeout 100 python tools/train_probe.py
 GPU_MAX_HW_QUEUES=4 TRAIN_VARIANTS=default,no_overlap timeout 100 python tools/train_probe.py) > gpurun_out/r2c23_train.log 2>&1
cat gpurun_out/r2c23_train.log
