#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
AB_REPS=2 bash tools/r5_ab_all.sh "GPK_EXTRA_MAX_WGS=224" "GPK_EXTRA_MAX_WGS=240" "GPK_EXTRA_MAX_WGS=192"
