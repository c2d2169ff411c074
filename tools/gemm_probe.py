"""GEMM throughput probe: C = alpha A B^T + beta C through the C-ABI for the shapes of the hot path."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

dev = ops.device()


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts)), float(np.min(ts))


g = torch.Generator(device="cpu").manual_seed(0)
shapes = [  # m, n, k, beta, c_lower
    (16384, 16384, 512, 1.0, False), (16384, 16384, 512, 0.0, False), (16384, 16384, 512, 1.0, True),
    (16384, 16384, 1024, 1.0, True), (8192, 8192, 4096, 0.0, False), (8192, 2048, 2048, 0.0, False),
    (8192, 256, 1792, 1.0, False), (8192, 128, 128, 0.0, False), (2048, 128, 128, 0.0, False),
    (1920, 128, 128, 1.0, True), (4096, 4096, 512, 1.0, True),
]
for (m, n, k, beta, low) in shapes:
    A = torch.randn((m, k), generator=g, dtype=torch.float64).to(dev)
    B = A if (m == n) else torch.randn((n, k), generator=g, dtype=torch.float64).to(dev)
    C = torch.zeros((m, n), dtype=torch.float64, device=dev)
    t, tmin = timeit(lambda: ops.gemm_nt(A, B, alpha=-1.0, beta=beta, C=C, c_lower=low))
    useful = (m * (m + 1) / 2 if low else m * n) * 2.0 * k
    print(json.dumps({"m": m, "n": n, "k": k, "beta": beta, "c_lower": low, "ms": round(tmin * 1e3, 4),
                      "tflops_useful": round(useful / tmin / 1e12, 2)}), flush=True)
    del A, B, C
