#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
AB_REPS=2 bash tools/r5_ab_all.sh "GPK_WHATIF_NO_EVR=1" "GPK_WHATIF_EVF_LATE=1" "GPK_WHATIF_NO_EVR=1 GPK_WHATIF_EVF_LATE=1"
