"""rocprofv3 workload: three SVGP steps at M = 2048, D = 16 (PROF_M / PROF_D in the environment override) with ROWS minibatch rows
(argv[1], default 1024)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
m, d = int(os.environ.get("PROF_M", "2048")), int(os.environ.get("PROF_D", "16"))
Z = ops.to_device(rng.normal(size=(m, d))); Xb = ops.to_device(rng.normal(size=(rows, d))); Yb = ops.to_device(rng.normal(size=(rows, 1)))
q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
ws = ops.svgp_elbo_workspace(m, rows, d, 1, False)
for _ in range(3):
    out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws)
torch.cuda.synchronize()
print("svgp", out.cpu().numpy())
