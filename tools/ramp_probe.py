"""Per-step wall-clock of the first steps of the SVGP Cm workload after a host-side pause: does the chip need a ramp?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpflow_amd import ops
dev = torch.device("cuda", 0)
X, Y, Z, q_mu, q_sqrt, ls = bench.make_inputs(200_000, 2048, 8, 4, dev)
b = 8192
ws = ops.svgp_elbo_workspace(2048, b, 8, 1, False)
out = torch.empty(2, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
def step(s):
    lo = (s % 20) * b
    ops.svgp_elbo_shard(Z, X[lo:lo + b], Y[lo:lo + b], q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws, out=out, info=info)
    return float(out.cpu()[0])
for pause in (0.0, 0.5, 3.0):
    time.sleep(pause)
    ts = []
    for s in range(160):
        t0 = time.perf_counter(); step(s); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(f"pause {pause:.1f}s: first 5 {np.round(ts[:5], 2)}  mean[5:25] {ts[5:25].mean():.3f}  mean[25:55] {ts[25:55].mean():.3f}  mean[55:105] {ts[55:105].mean():.3f}  mean[105:160] {ts[105:160].mean():.3f}", flush=True)
