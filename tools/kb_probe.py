"""K builder (full symmetric 16384^2, D = 8): time per launch and write rate; bitwise symmetry check."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops
n, d = 16384, 8
rng = np.random.default_rng(2)
X = ops.to_device(rng.normal(size=(n, d)))
ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
K = torch.empty((n, n), dtype=torch.float64, device=X.device)
ts = []
for i in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.kernel_matrix(X, None, variance=1.0, lengthscales=ls, diag_add=0.1, out=K); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
t = min(ts[2:])
sym = bool(torch.equal(K[:4096, :4096], K[:4096, :4096].t()))
print("kb ms %.4f  GB/s %.0f  frac_of_8TB/s %.3f  symmetric %s  tag %s" % (t * 1e3, (n * n * 8 + n * d * 8) / t / 1e9, (n * n * 8 + n * d * 8) / t / 8e12, sym, os.environ.get("TAG", "")))
