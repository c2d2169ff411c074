"""Under-filled projections (few rows): value against NumPy and time per launch.  Run with the A/B library and
GPK_PROJ_HALF_TILE_BELOW=<pairs> to switch the 64-row, unpaired tiles on."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
for rows, m, P in [(1024, 2048, 1), (300, 1024, 2), (2048, 2048, 1), (4096, 2048, 1), (1024, 1024, 4), (8192, 1024, 1), (8192, 2048, 1)]:
    At = rng.normal(size=(rows, m))
    q = rng.normal(size=(P, m, m))
    LqT = ops.transpose(ops.to_device(q), mode=1)
    A = ops.to_device(At)
    ssq = ops.project(A, LqT)
    ref = np.stack([((At @ np.tril(q[p])) ** 2).sum(1) for p in range(P)])
    err = np.abs(ssq.cpu().numpy() - ref).max() / np.abs(ref).max()
    for _ in range(3):
        ops.project(A, LqT)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.project(A, LqT)
    e1.record(); torch.cuda.synchronize()
    print(f"rows={rows} m={m} P={P}: rel err {err:.1e}, {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call", flush=True)
