"""Workload for rocprofv3 --kernel-trace --stats: a few SVGP steps (Cm) and one N=16384 GPR LML."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "r1_ref"))
from gpflow_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
rng = np.random.default_rng(0)
dev = ops.device()
if which in ("svgp", "both"):
    m, B, d = 2048, 8192, 8
    Z = ops.to_device(rng.normal(size=(m, d)))
    Xb = ops.to_device(rng.normal(size=(B, d)))
    Yb = ops.to_device(rng.normal(size=(B, 1)))
    q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
    q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    ws = ops.svgp_elbo_workspace(m, B, d, 1, False)
    for _ in range(3):
        out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls,
                                        noise_variance=0.1, jitter=1e-6, ws=ws)
    torch.cuda.synchronize()
    print("svgp", out.cpu().numpy())
if which in ("gpr", "both"):
    n, d = 16384, 8
    X = ops.to_device(rng.normal(size=(n, d)))
    Y = ops.to_device(rng.normal(size=(n, 1)))
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    for _ in range(2):
        out, info = ops.gpr_lml(X, Y, variance=1.0, lengthscales=ls, noise_variance=0.1)
    torch.cuda.synchronize()
    print("gpr", out.cpu().numpy(), info.cpu().numpy())
