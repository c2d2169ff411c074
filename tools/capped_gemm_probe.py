"""Time the extra-row update GEMM of the SVGP step (8192 x {1536,1024,512} -= [8192 x 512] [.. x 512]^T) alone, under the A/B
library's GPK_GEMM_NT_MAX_WGS cap.  usage: GPK_LIBRARY=.../libgpk_exp.so GPK_GEMM_NT_MAX_WGS=224 python tools/capped_gemm_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
rows, k = 8192, 512
A = ops.to_device(rng.normal(size=(rows, k)))
res = []
for n in (1536, 1024, 512):
    B = ops.to_device(rng.normal(size=(n, k)))
    C = ops.to_device(rng.normal(size=(rows, n)))
    for _ in range(3):
        ops.gemm_nt(A, B, alpha=-1.0, beta=1.0, C=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); ops.gemm_nt(A, B, alpha=-1.0, beta=1.0, C=C); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = float(np.median(ts))
    res.append("n=%d %.0f us %.1f TFLOP/s" % (n, us, 2.0 * rows * n * k / us / 1e6))
print("cap=%s excl=%s tail=%s : " % (os.environ.get("GPK_GEMM_NT_MAX_WGS", "0"), os.environ.get("GPK_CAP_EXCL_LDS_KB", "84"),
                                      os.environ.get("GPK_TAIL_SPLIT_CAPPED", "1")) + " | ".join(res))
