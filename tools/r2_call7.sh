#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
export GPK_X=$root/gpflow_amd/libgpk_exp.so
OLD="GPK_SOFT_RESERVE=0 GPK_GROUP_INVERSE=0 GPK_EXTRA_MAX_WGS=320"
tools/ab.sh "$OLD" "GPK_SOFT_RESERVE=0" "GPK_GROUP_INVERSE=0" "GPK_STREAM_PROJ=1" "GPK_STREAM_PROJ=1 GPK_SOFT_RESERVED_CUS=16" "GPK_SOFT_RESERVED_CUS=16" "GPK_SOFT_RESERVED_CUS=48" "GPU_MAX_HW_QUEUES=4" "GPK_STREAM_PROJ=1 GPU_MAX_HW_QUEUES=4" > gpurun_out/r2c7_ab.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-extras > gpurun_out/r2c7_bench_gpr.json 2> gpurun_out/r2c7_bench_gpr.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $root/gpurun_out/r2c7_prof -o svgp -- python $root/tools/prof_run.py svgp > $root/gpurun_out/r2c7_prof.log 2>&1
GPK_LIBRARY=$GPK_X GPK_STREAM_PROJ=1 rocprofv3 --kernel-trace -d $root/gpurun_out/r2c7_prof_s -o svgp -- python $root/tools/prof_run.py svgp > $root/gpurun_out/r2c7_prof_s.log 2>&1
cd $root
for d in r2c7_prof r2c7_prof_s; do
  db=$(find gpurun_out/$d -name "*.db" | head -1)
  python tools/timeline.py $db rbf_kernel 4 170 > gpurun_out/${d}_timeline.txt 2>&1
  rm -rf gpurun_out/$d
done
cat gpurun_out/r2c7_ab.log; python -c "
import json; d=json.load(open('gpurun_out/r2c7_bench_gpr.json')); g=d['gpr_cholesky']; print('GPR ms', g['ms_total'], 'trail', g['trailing_update_roofline']['achieved'], 'pred ms', g['predict']['ms_total'], g['predict']['cached_posterior_ms'])"
