#!/bin/bash
# A/B of the GPR N=16384 leg only (bench.py gpr_cholesky), same box:  tools/ab_gpr.sh "VAR=1" ...
for cfg in "" "$@"; do
  env ABCFG="$cfg" $cfg python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench, torch
from gpflow_amd import _lib, ops
r = bench.gpr_cholesky_leg(ops, _lib.load(), torch.device("cuda", 0))
print("cfg=[%s] gpr_ms=%.2f tflops=%.1f" % (os.environ.get("ABCFG", ""), r["ms_total"], r["cholesky_gflops_incl_build_and_tail"] / 1e3))
PY
done
