"""Time GPR LML value+gradient at N=16384, D=8 (gradients.gpr_lml_and_grad)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import gradients
N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 8
g = torch.Generator().manual_seed(2)
X = torch.randn((N, D), generator=g, dtype=torch.float64).cuda()
Y = torch.sin(X.sum(1, keepdim=True)) + 0.1 * torch.randn((N, 1), dtype=torch.float64, device="cuda")
kw = dict(variance=1.0, lengthscales=np.sqrt(D) * (0.8 + 0.05 * np.arange(D)), noise_variance=0.1)
F, gr, info = gradients.gpr_lml_and_grad(X, Y, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    F, gr, info = gradients.gpr_lml_and_grad(X, Y, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("N=%d value+grad ms: %.1f  lml=%.6f info=%d dvar=%.6f" % (N, dt * 1e3, float(F.cpu()[0]), int(info.cpu()[0]), float(gr["variance"].cpu()[0])))
