#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
X="GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so"
bash tools/prof_timeline.sh r3c12_cap192 $X GPK_EXTRA_MAX_WGS=192
bash tools/prof_timeline.sh r3c12_cap256 $X GPK_EXTRA_MAX_WGS=256
