#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
X="GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so"
( env $X GPK_GEMM_W8=2 timeout 300 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "gemm or potrf or trsm or project" 2>&1 | tail -4 ) > $out/r3w8_pytest.log 2>&1
( env $X PROBE_TAG=w4 timeout 120 python tools/gemm_probe_x.py; env $X GPK_GEMM_W8=2 PROBE_TAG=w8 timeout 120 python tools/gemm_probe_x.py ) > $out/r3w8_probe.log 2>&1
