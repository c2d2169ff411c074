#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
AB_REPS=2 bash tools/r5_ab_all.sh "GPK_PROJ_SMALL_TILE_BELOW=300" "GPK_PROJ_SMALL_TILE_BELOW=520"
