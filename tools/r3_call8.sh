#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
X="GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so"
export AB_REPS=2
bash tools/ab.sh \
  "GPK_UMASK=1" \
  "GPK_UMASK=1 GPK_RESERVED_CUS=32" \
  "GPK_UMASK=1 GPK_RESERVED_CUS=64" \
  "GPK_UMASK=1 GPK_RESERVED_CUS=96" \
  "GPK_UMASK=1 GPK_RESERVED_CUS=64 GPK_UMASK_CAP=384" \
  "GPK_UMASK=1 GPK_RESERVED_CUS=48" \
  "GPK_UMASK=1 GPK_RESERVED_CUS=16" \
  "GPK_EXTRA_MAX_WGS=384" \
  > $out/r3c8_ab.log 2>&1
bash tools/prof_timeline.sh r3c8_umask64 $X GPK_UMASK=1 GPK_RESERVED_CUS=64
