#!/bin/bash
# kernel-trace statistics of the bench command whose dominant-kernel launches all have the Cm shape (--no-other)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r04_bench -o bench -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other > $root/gpurun_out/r04_bench_under_rocprof.json 2> $root/gpurun_out/r04_bench.err
cd $root
python tools/rocpd_summary.py $(find gpurun_out/r04_bench -name "*.db" | head -1) > gpurun_out/r04_bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/r04_bench
