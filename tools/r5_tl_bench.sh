#!/bin/bash
# kernel timeline of one step of the default bench.py workload (host waits for every step's scalars, as the driver's run does):
#   tools/r5_tl_bench.sh <tag> [ENV=...]...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 250 rocprofv3 --kernel-trace -d $root/gpurun_out/${tag}_tlb -o tl -- python $root/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other --no-gpr --no-train --no-extras > $root/gpurun_out/${tag}_tlb.log 2>&1
cd $root
db=$(find gpurun_out/${tag}_tlb -name "*.db" | head -1)
python tools/timeline.py $db rbf_kernel 20 130 > gpurun_out/${tag}_bench_timeline.txt 2>&1
rm -rf gpurun_out/${tag}_tlb
