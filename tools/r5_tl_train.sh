#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $root/gpurun_out/r05_train_tl -o tl -- python $root/tools/prof_run.py train > $root/gpurun_out/r05_train_tl.log 2>&1 )
db=$(find gpurun_out/r05_train_tl -name "*.db" | head -1)
python tools/timeline.py $db rbf_kernel ${OCC:-8} 400 > gpurun_out/r05_train_timeline.txt 2>&1
rm -rf gpurun_out/r05_train_tl
wc -l gpurun_out/r05_train_timeline.txt
