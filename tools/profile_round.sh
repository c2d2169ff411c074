#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace of the default bench.py command + PMC passes on the short profiling
# workload.  Raw rocpd DBs land in gpurun_out/<tag>_*/ ; summaries are made here by tools/rocpd_summary.py and
# tools/pmc_summary.py (text files next to them) and copied to profiles/ by hand.
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_bench -o bench -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other > $root/gpurun_out/${tag}_bench_under_rocprof.json 2> $root/gpurun_out/${tag}_bench.err
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $root/gpurun_out/${tag}_pmc_fetch -o pmc -- python $root/tools/prof_run.py both > $root/gpurun_out/${tag}_pmc_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $root/gpurun_out/${tag}_pmc_write -o pmc -- python $root/tools/prof_run.py both > $root/gpurun_out/${tag}_pmc_write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $root/gpurun_out/${tag}_pmc_mfma -o pmc -- python $root/tools/prof_run.py both > $root/gpurun_out/${tag}_pmc_mfma.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $root/gpurun_out/${tag}_pmc_kb_write -o pmc -- python $root/tools/kb_probe.py > $root/gpurun_out/${tag}_pmc_kb_write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $root/gpurun_out/${tag}_pmc_kb_fetch -o pmc -- python $root/tools/kb_probe.py > $root/gpurun_out/${tag}_pmc_kb_fetch.log 2>&1
cd $root
python tools/rocpd_summary.py $(find gpurun_out/${tag}_bench -name "*.db" | head -1) > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
for k in fetch write mfma kb_write kb_fetch; do
  python tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_$k -name "*.db" | head -1) > gpurun_out/${tag}_pmc_$k.txt 2>&1
done
db=$(find gpurun_out/${tag}_bench -name "*.db" | head -1)
python tools/timeline.py $db rbf_kernel 20 130 > gpurun_out/${tag}_step_timeline.txt 2>&1
rm -rf gpurun_out/${tag}_bench gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_mfma gpurun_out/${tag}_pmc_kb_write gpurun_out/${tag}_pmc_kb_fetch
ls -la gpurun_out/${tag}_*
tail -c 600 gpurun_out/${tag}_bench_under_rocprof.json
