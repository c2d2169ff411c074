"""Per-rank step time of the C4 strong-scaling workload (M = 2048, D = 16, one 8192-row minibatch split over G ranks)
measured on ONE GPU for G = 1, 2, 4, 8: the shard of rank 0, everything a rank does per step except the 8-byte
all-reduce.  Gives the strong-scaling ceiling the replicated latency chain allows."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpflow_amd import ops
dev = torch.device("cuda", 0)
n_data, m, d, rows_global, _, seed = bench.WORKLOADS["c4-strong"]
X, Y, Z, q_mu, q_sqrt, ls = bench.make_inputs(1_000_000, m, d, seed, dev)   # (1e6 rows suffice: only slices are touched)
h = torch.empty(2, dtype=torch.float64).pin_memory()
base = None
for G in (1, 2, 4, 8):
    b = rows_global // G
    ws = ops.svgp_elbo_workspace(m, b, d, 1, False)
    out = torch.empty(2, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    def step(s):
        lo = (s % 100) * rows_global
        ops.svgp_elbo_shard(Z, X[lo:lo + b], Y[lo:lo + b], q_mu, q_sqrt, variance=1.0, lengthscales=ls,
                            noise_variance=0.1, jitter=1e-6, ws=ws, out=out, info=info)
        h.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    for s in range(5): step(s)
    t0 = time.perf_counter()
    for s in range(50): step(s)
    dt = (time.perf_counter() - t0) / 50
    base = base or dt
    print("G=%d rows/rank=%5d  ms/step %.3f  global steps/s %.1f  speed-up over G=1 %.2fx (ceiling: no all-reduce latency)"
          % (G, b, dt * 1e3, 1.0 / dt, base / dt), flush=True)
