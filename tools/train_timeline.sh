#!/bin/bash
# kernel timeline of one forward + reverse SVGP evaluation at Cm (gradients.svgp_elbo_and_grad): tools/train_timeline.sh <tag> [ENV=...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 200 rocprofv3 --kernel-trace -d $root/gpurun_out/${tag}_tl -o tl -- python $root/tools/prof_run.py train > $root/gpurun_out/${tag}_tl.log 2>&1
cd $root
db=$(find gpurun_out/${tag}_tl -name "*.db" | head -1)
python tools/timeline.py $db rbf_kernel 9 260 > gpurun_out/${tag}_timeline.txt 2>&1
rm -rf gpurun_out/${tag}_tl
