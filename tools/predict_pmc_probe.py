"""rocprofv3 --pmc workload: THREE fused GPR.predict_f calls at config C2 (N = 16384, D = 8, T = 4096) and nothing else, so that the sum
of a counter over every dispatch of the process / 3 is the per-call figure (bench.py -> gpr_cholesky.predict.roofline.traffic)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpflow_amd as gpflow
N, D, T = 16384, 8, 4096
rng = np.random.default_rng(2)
X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, 1))
Xn = gpflow.ops.to_device(np.random.default_rng(3).normal(size=(T, D)))
ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D))
m = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=ls), noise_variance=0.1)
for _ in range(3):
    mu, var = m.predict_f(Xn)
torch.cuda.synchronize()
print("predict calls: 3", float(mu.abs().max()), float(var.min()))
