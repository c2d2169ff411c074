#!/bin/bash
# kernel timeline of one evaluation of a tools/prof_run.py workload:  tools/mode_timeline.sh <tag> <mode> <marker kernel> <occurrence> <rows> [ENV=...]
tag=$1; mode=$2; marker=$3; occ=$4; nrows=$5; shift 5
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 200 rocprofv3 --kernel-trace -d $root/gpurun_out/${tag}_tl -o tl -- python $root/tools/prof_run.py $mode > $root/gpurun_out/${tag}_tl.log 2>&1
cd $root
db=$(find gpurun_out/${tag}_tl -name "*.db" | head -1)
python tools/timeline.py $db $marker $occ $nrows > gpurun_out/${tag}_timeline.txt 2>&1
rm -rf gpurun_out/${tag}_tl
