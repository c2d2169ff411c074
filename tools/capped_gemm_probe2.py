"""Per-round time of the capped persistent GEMM against K and beta: 8192 x 1792 output = 896 tiles = 4 whole rounds of 224
workgroups.  usage: GPK_LIBRARY=.../libgpk_exp.so GPK_GEMM_NT_MAX_WGS=224 GPK_TAIL_SPLIT_CAPPED=0 python tools/capped_gemm_probe2.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
rows, n = 8192, 1792
res = []
for k in (256, 512, 1024, 2048):
    A = ops.to_device(rng.normal(size=(rows, k)))
    B = ops.to_device(rng.normal(size=(n, k)))
    C = ops.to_device(rng.normal(size=(rows, n)))
    for beta in (0.0, 1.0):
        for _ in range(3):
            ops.gemm_nt(A, B, alpha=-1.0, beta=beta, C=C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(8):
            e0.record(); ops.gemm_nt(A, B, alpha=-1.0, beta=beta, C=C); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        us = float(np.median(ts))
        res.append("K=%d beta=%g: %.0f us (%.1f / round, MFMA floor %.1f)" % (k, beta, us, us / 4, 2.0 * 128 * 128 * k / 307.2e3))
print("cap=%s pc=%s\n  " % (os.environ.get("GPK_GEMM_NT_MAX_WGS", "0"), os.environ.get("GPK_CAP_PREFETCH_C", "1")) + "\n  ".join(res))
