#!/bin/bash
# Round-5 evidence (one gpurun call): profile_round.sh r05 (kernel stats of the bench command, PMC passes, step timeline) + the PMC
# passes of the fused GPR predict call (whole-call traffic) + timelines of the chain-bound regime with and without the chain flags.
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/r05_pred_$c -o pmc -- python $root/tools/predict_pmc_probe.py > $root/gpurun_out/r05_pred_$c.log 2>&1 )
  python tools/pmc_total.py $(find gpurun_out/r05_pred_$c -name "*.db" | head -1) 3 > gpurun_out/r05_pmc_predict_$c.txt 2>&1
  rm -rf gpurun_out/r05_pred_$c
done
TL_LINES=70 bash tools/r5_tl_rows.sh r05_rows1024_flags 1024 "GPK_CHAIN_FLAGS=1" > /dev/null 2>&1
TL_LINES=70 bash tools/r5_tl_rows.sh r05_rows1024_events 1024 "GPK_CHAIN_FLAGS=0" > /dev/null 2>&1
ls -la gpurun_out/r05_*
cat gpurun_out/r05_pmc_predict_*.txt
