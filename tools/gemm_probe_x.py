"""GEMM probe for the extra-row-stream shapes of the SVGP step (8192 rows): alone on the chip, per launch."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

dev = ops.device()


def timeit(fn, reps=7, warm=2):
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
shapes = [  # m, n, k, beta, b_tri
    (8192, 512, 512, 1.0, 0), (8192, 1024, 512, 1.0, 0), (8192, 1536, 512, 1.0, 0), (8192, 2048, 512, 1.0, 0),
    (8192, 512, 512, 0.0, 2), (8192, 512, 512, 0.0, 0), (8192, 512, 1536, 1.0, 0), (8192, 512, 1024, 1.0, 0),
    (8192, 256, 256, 1.0, 0), (8192, 128, 128, 1.0, 0), (8192, 384, 128, 1.0, 0), (4096, 512, 512, 1.0, 0),
    (8192, 1536, 1536, 0.0, 1), (8192, 2048, 2048, 0.0, 1),
]
tag = os.environ.get("PROBE_TAG", "")
for (m, n, k, beta, tri) in shapes:
    A = torch.randn((m, k), generator=g, dtype=torch.float64).to(dev)
    B = torch.randn((n, k), generator=g, dtype=torch.float64).to(dev)
    if tri == 2:
        B = torch.tril(B)
    if tri == 1:
        B = torch.triu(B)
    C = torch.zeros((m, n), dtype=torch.float64, device=dev)
    t, tmin = timeit(lambda: ops.gemm_nt(A, B, alpha=-1.0 if beta else 1.0, beta=beta, C=C, b_tri=tri))
    useful = 2.0 * m * k * n * (0.5 if tri else 1.0)
    print(json.dumps({"tag": tag, "m": m, "n": n, "k": k, "beta": beta, "b_tri": tri, "us_min": round(tmin * 1e6, 1),
                      "us_med": round(t * 1e6, 1), "tflops": round(useful / tmin / 1e12, 2)}), flush=True)
    del A, B, C
