"""In-group solve alone: gpk_trsm(trans = 0) of `rows` right-hand sides against a 512 x 512 factor (ONE fused group of four leaf blocks).
usage: GPK_LIBRARY=.../libgpk_exp.so GPK_GROUP_SOLVE_V2=0|1 python tools/group_solve_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
n = 512
Z = rng.normal(size=(n, 4))
K = np.exp(-0.5 * ((Z[:, None] - Z[None]) ** 2).sum(-1) / 4.0) + 0.1 * np.eye(n)
L = ops.to_device(np.linalg.cholesky(K))
invd = ops.trtri_blocks(L)
res = []
for rows in (1024, 2048, 4096, 8192):
    B0 = ops.to_device(rng.normal(size=(rows, n)))
    B = B0.clone()
    for _ in range(3):
        B.copy_(B0); ops.trsm_(B, L, invd, trans=0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        B.copy_(B0); torch.cuda.synchronize()
        e0.record(); ops.trsm_(B, L, invd, trans=0); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    res.append("rows=%d %.1f us" % (rows, float(np.median(ts))))
print("v2=%s: " % os.environ.get("GPK_GROUP_SOLVE_V2", "1") + " | ".join(res))
