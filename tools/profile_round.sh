#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace of the default bench.py command + PMC passes on the short
# profiling workload.  Raw rocpd DBs land in gpurun_out/<tag>_*/ ; summaries are made by tools/rocpd_summary.py
# and tools/pmc_summary.py and copied to profiles/ by hand.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_bench -o bench -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $root/gpurun_out/${tag}_bench.json 2> $root/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $root/gpurun_out/${tag}_pmc_fetch -o pmc -- python $root/tools/prof_run.py both > $root/gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $root/gpurun_out/${tag}_pmc_write -o pmc -- python $root/tools/prof_run.py both > $root/gpurun_out/${tag}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $root/gpurun_out/${tag}_pmc_mfma -o pmc -- python $root/tools/prof_run.py both > $root/gpurun_out/${tag}_pmc_mfma.log 2>&1
ls -la $root/gpurun_out/${tag}_*/
tail -2 $root/gpurun_out/${tag}_bench.json | cut -c1-600
