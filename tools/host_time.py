"""Host time inside gpk_svgp_elbo_shard vs wall time per step (is the step host-bound?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpflow_amd import ops
dev = torch.device("cuda", 0)
X, Y, Z, q_mu, q_sqrt, ls = bench.make_inputs(0, dev)
ws = ops.svgp_elbo_workspace(2048, 8192, 8, 1, False)
out = torch.empty(2, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
def call(s):
    lo = s * 8192
    t0 = time.perf_counter()
    ops.svgp_elbo_shard(Z, X[lo:lo + 8192], Y[lo:lo + 8192], q_mu, q_sqrt, variance=1.0, lengthscales=ls,
                        noise_variance=0.1, jitter=1e-6, ws=ws, out=out, info=info)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t0
for s in range(5): call(s)
h, w = zip(*[call(5 + s) for s in range(30)])
print("host enqueue ms: median %.3f  | wall per step ms: median %.3f" % (np.median(h) * 1e3, np.median(w) * 1e3))
