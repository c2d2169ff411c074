#!/bin/bash
# FETCH_SIZE of the kernels of three SVGP steps (tools/prof_run.py svgp):  tools/pmc_fetch_svgp.sh <tag> [ENV=...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $root/gpurun_out/${tag}_pmc -o pmc -- python $root/tools/prof_run.py svgp > $root/gpurun_out/${tag}_pmc.log 2>&1
cd $root
python tools/pmc_summary.py $(find gpurun_out/${tag}_pmc -name "*.db" | head -1) > gpurun_out/${tag}_pmc_fetch.txt 2>&1
rm -rf gpurun_out/${tag}_pmc
