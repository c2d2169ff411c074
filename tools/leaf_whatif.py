"""What would a faster leaf buy?  Times the SVGP step (Cm) of the A/B library with GPK_LEAF_FAKE_US (a stand-in leaf of
a given duration; results meaningless, launch structure unchanged)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpflow_amd import ops
dev = torch.device("cuda", 0)
n_data, m, d, b, _, seed = bench.WORKLOADS["cm"]
X, Y, Z, q_mu, q_sqrt, ls = bench.make_inputs(n_data, m, d, seed, dev)
ws = ops.svgp_elbo_workspace(m, b, d, 1, False)
out = torch.empty(2, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
def step(s):
    lo = (s % 100) * b
    ops.svgp_elbo_shard(Z, X[lo:lo + b], Y[lo:lo + b], q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1,
                        jitter=1e-6, ws=ws, out=out, info=info)
    torch.cuda.current_stream().synchronize()
for s in range(5): step(s)
t0 = time.perf_counter()
for s in range(40): step(s)
dt = (time.perf_counter() - t0) / 40
print("leaf_fake_us=%s  ms/step %.3f" % (os.environ.get("GPK_LEAF_FAKE_US", "real"), dt * 1e3))
