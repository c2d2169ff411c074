#!/bin/bash
# final evidence of round 2 on the final product code: full GPU suite, default bench, profiles, training probe
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
rm -f gpurun_out/r02_kb_ab.log
bash tools/r2_evidence.sh
(timeout 200 python tools/train_probe.py table gpr) > gpurun_out/r02_train_probe.log 2>&1
cat gpurun_out/r02_train_probe.log
