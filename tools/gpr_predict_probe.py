"""GPR config C2 predict at full size (N=16384, D=8, T=4096): fused route (trapezoid with T extra rows) vs cached
posterior (Lm + blocked trsm); prints agreement and times."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpflow_amd as gpflow
N, D, T = 16384, 8, 4096
rng = np.random.default_rng(2)
X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, 1))
Xn = np.random.default_rng(3).normal(size=(T, D))
ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D))
m = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=ls), noise_variance=0.1)
Xd = gpflow.ops.to_device(Xn)
mu, var = m.predict_f(Xd); torch.cuda.synchronize()
t0 = time.perf_counter(); mu, var = m.predict_f(Xd); torch.cuda.synchronize(); t_fused = time.perf_counter() - t0
post = m.posterior(); torch.cuda.synchronize()
mu2, var2 = post.predict_f(Xd); torch.cuda.synchronize()
t0 = time.perf_counter(); mu2, var2 = post.predict_f(Xd); torch.cuda.synchronize(); t_cached = time.perf_counter() - t0
print("fused ms %.1f cached ms %.1f" % (t_fused * 1e3, t_cached * 1e3))
print("mean maxdiff %.3e (max |mean| %.3f)  var maxdiff %.3e (min var %.3e max var %.3e)" % (
    float((mu - mu2).abs().max()), float(mu.abs().max()), float((var - var2).abs().max()), float(var.min()), float(var.max())))
print("done")
