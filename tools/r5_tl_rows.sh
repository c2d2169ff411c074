#!/bin/bash
# kernel timeline of one M = 2048 step with few rows (the chain-bound regime):  tools/r5_tl_rows.sh tag rows "ENV=.. ENV=.." ...
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
while [ $# -gt 0 ]; do
  tag=$1; rows=$2; cfg=$3; shift; shift; shift
  ( cd /tmp && export TMPDIR=/tmp && env GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so $cfg timeout 200 rocprofv3 --kernel-trace -d $root/gpurun_out/${tag}_tl -o tl -- python $root/tools/prof_run_rows.py $rows > $root/gpurun_out/${tag}_tl.log 2>&1 )
  db=$(find gpurun_out/${tag}_tl -name "*.db" | head -1)
  python tools/timeline.py $db rbf_kernel 4 120 > gpurun_out/${tag}_timeline.txt 2>&1
  rm -rf gpurun_out/${tag}_tl
  head -${TL_LINES:-60} gpurun_out/${tag}_timeline.txt
done
