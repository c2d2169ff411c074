#!/bin/bash
# timelines of one Cm step under A/B-library settings:  tools/r5_tl.sh tag "ENV=.. ENV=.." [tag2 "ENV.."]...
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
while [ $# -gt 0 ]; do
  tag=$1; cfg=$2; shift; shift
  bash tools/prof_timeline.sh $tag GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so $cfg
  head -100 gpurun_out/${tag}_timeline.txt
done
