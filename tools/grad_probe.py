"""Time one forward+backward of the SVGP step at the benchmark size (M=2048, B=8192, D=8, P=1)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import gradients, ops
M, B, D, P = 2048, 8192, 8, 1
g = torch.Generator().manual_seed(0)
X = torch.randn((B, D), generator=g, dtype=torch.float64).cuda()
Y = torch.sin(X.sum(1, keepdim=True))
Z = X[:M].clone() + 0.01
q_mu = (0.1 * torch.randn((M, P), generator=g, dtype=torch.float64)).cuda()
q_sqrt = (torch.tril(0.05 * torch.randn((P, M, M), generator=g, dtype=torch.float64)) + 0.5 * torch.eye(M, dtype=torch.float64)).cuda()
kw = dict(variance=1.0, lengthscales=np.sqrt(D) * (0.8 + 0.05 * np.arange(D)), noise_variance=0.1, jitter=1e-6, scale=1e6 / B)
for _ in range(3):
    F, gr, info = gradients.svgp_elbo_and_grad(Z, X, Y, q_mu, q_sqrt, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    F, gr, info = gradients.svgp_elbo_and_grad(Z, X, Y, q_mu, q_sqrt, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("forward+backward ms: %.3f  (%.1f steps/s)  F=%.6f info=%d" % (dt * 1e3, 1 / dt, float(F.cpu()[0]), int(info.cpu()[0])))
