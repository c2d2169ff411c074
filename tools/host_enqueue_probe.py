"""How long does the HOST take to enqueue one fused SVGP step (Cm, C3), against the step's GPU time?  Run on the GPU box.
Prints per workload: host time of the C-ABI call (no synchronisation), and the synchronised step time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
for name, m, B in (("cm", 2048, 8192), ("c3", 1024, 8192), ("rows1024", 2048, 1024)):
    d = 8
    Z = ops.to_device(rng.normal(size=(m, d)))
    Xb = ops.to_device(rng.normal(size=(B, d)))
    Yb = ops.to_device(rng.normal(size=(B, 1)))
    q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
    q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    ws = ops.svgp_elbo_workspace(m, B, d, 1, False)
    out = torch.empty(2, dtype=torch.float64, device=ops.device())
    info = torch.zeros(1, dtype=torch.int32, device=ops.device())
    host, total = [], []
    for it in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws,
                            out=out, info=info)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it >= 10:
            host.append((t1 - t0) * 1e6)
            total.append((t2 - t0) * 1e6)
    print("%-9s host enqueue %.0f us (min %.0f)   synchronised step %.0f us (min %.0f)" %
          (name, np.median(host), min(host), np.median(total), min(total)))
