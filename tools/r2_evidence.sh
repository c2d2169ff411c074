#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
(time timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > gpurun_out/r02_gpu_tests.log 2>&1
tail -n 6 gpurun_out/r02_gpu_tests.log
export X=$root/gpflow_amd/libgpk_exp.so
for cfg in "TAG=product" "GPK_LIBRARY=$X TAG=lds_mirror_off GPK_RBF_LDS_MIRROR=0" "GPK_LIBRARY=$X TAG=nt GPK_RBF_NT_STORE=1" "GPK_LIBRARY=$X TAG=nt_nomirror GPK_RBF_NT_STORE=1 GPK_RBF_LDS_MIRROR=0"; do
  env $cfg timeout 60 python tools/kb_probe.py 2>/dev/null | grep "kb ms" >> gpurun_out/r02_kb_ab.log
done
cat gpurun_out/r02_kb_ab.log
timeout 400 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
tail -c 1500 gpurun_out/r02_bench.json
bash tools/profile_round.sh r02 > gpurun_out/r02_profile_round.log 2>&1
tail -n 12 gpurun_out/r02_profile_round.log
