"""Does one more active stream in the process (what RCCL brings: an internal stream that waits for the compute stream,
runs a tiny kernel and is waited for) disturb the library's stream-to-queue layout?  SVGP step (Cm) with and without a
stand-in "collective" per step on a torch side stream created (a) before and (b) after the library's first call."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpflow_amd import ops
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "none"   # none | before | after
side = torch.cuda.Stream() if mode == "before" else None
n_data, m, d, b, _, seed = bench.WORKLOADS["cm"]
X, Y, Z, q_mu, q_sqrt, ls = bench.make_inputs(n_data, m, d, seed, dev)
ws = ops.svgp_elbo_workspace(m, b, d, 1, False)
out = torch.empty(2, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
h = torch.empty(2, dtype=torch.float64).pin_memory()
def step(s):
    lo = (s % 100) * b
    ops.svgp_elbo_shard(Z, X[lo:lo + b], Y[lo:lo + b], q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1,
                        jitter=1e-6, ws=ws, out=out, info=info)
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out[0:1].mul_(1.0)          # stand-in for the 8-byte all-reduce
        torch.cuda.current_stream().wait_stream(side)
    h.copy_(out, non_blocking=True)
    torch.cuda.current_stream().synchronize()
step(0)
if mode == "after":
    side = torch.cuda.Stream()
for s in range(5): step(s)
t0 = time.perf_counter()
for s in range(50): step(s)
dt = (time.perf_counter() - t0) / 50
print("side stream: %-6s  ms/step %.3f  steps/s %.1f" % (mode, dt * 1e3, 1.0 / dt))
