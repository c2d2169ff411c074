"""Does replaying the SVGP ELBO step (Cm) as ONE hipGraph change its device time?  (torch.cuda.CUDAGraph capture around
ops.svgp_elbo_shard: the library's internal streams fork from / join to the capturing stream through events.)"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
m, B, d = 2048, 8192, 8
Z = ops.to_device(rng.normal(size=(m, d)))
Xb = ops.to_device(rng.normal(size=(B, d)))
Yb = ops.to_device(rng.normal(size=(B, 1)))
q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
ws = ops.svgp_elbo_workspace(m, B, d, 1, False)
kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws)


def step():
    return ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, **kw)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    out, info = step()
ref = out.cpu().numpy().copy()
print("eager ms/step", [round(timed(step, 40), 4) for _ in range(3)], flush=True)
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(side):
        step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            gout, ginfo = step()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("graph result equal:", bool((gout.cpu().numpy() == ref).all()), gout.cpu().numpy(), ref, flush=True)
    print("graph ms/step", [round(timed(g.replay, 40), 4) for _ in range(3)], flush=True)
    print("eager again  ", [round(timed(step, 40), 4) for _ in range(2)], flush=True)
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:600], flush=True)
