"""Config C5 (multi-output SVGP, 4 latent GPs, M = 1024, B = 8192, D = 8) through the model surface: ELBO steps/s for
(i) SharedIndependent + shared inducing points (one Cholesky, P-batched projection) and (ii) SeparateIndependent (batched
[4, M, M] Cholesky + batched solves), whitened and un-whitened.  Minibatches are device tensors."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpflow_amd as gp
from gpflow_amd import ops
rng = np.random.default_rng(8)
M, B, D, P = 1024, 8192, 8, 4
Xall = ops.to_device(rng.normal(size=(8 * B, D)))
Yall = ops.to_device(rng.normal(size=(8 * B, P)))
Z = rng.normal(size=(M, D))
q_mu = 0.1 * rng.normal(size=(M, P))
q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(M, M))) + 0.5 * np.eye(M) for _ in range(P)])
ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D))
iv = lambda: gp.inducing_variables.SharedIndependentInducingVariables(gp.inducing_variables.InducingPoints(Z))
def models():
    yield "C5(i)  SharedIndependent, whitened", gp.models.SVGP(
        gp.kernels.SharedIndependent(gp.kernels.SquaredExponential(variance=1.0, lengthscales=ls), output_dim=P),
        gp.likelihoods.Gaussian(0.1), iv(), q_mu=q_mu, q_sqrt=q_sqrt, num_latent_gps=P, num_data=1_000_000)
    for wh in (True, False):
        kern = gp.kernels.SeparateIndependent([gp.kernels.SquaredExponential(variance=v, lengthscales=l)
                                               for v, l in zip([1.0, 0.8, 1.2, 0.9], [2.4, 2.8, 3.2, 3.6])])
        yield "C5(ii) SeparateIndependent, whiten=%s" % wh, gp.models.SVGP(
            kern, gp.likelihoods.Gaussian(0.1), iv(), q_mu=q_mu, q_sqrt=q_sqrt, num_latent_gps=P, whiten=wh, num_data=1_000_000)
for name, m in models():
    def step(s):
        lo = (s % 8) * B
        return float(m.elbo((Xall[lo:lo + B], Yall[lo:lo + B])))
    for s in range(3): v = step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(20): v = step(s)
    dt = (time.perf_counter() - t0) / 20
    print("%-42s ms/step %.3f  steps/s %.1f  last elbo %.6f" % (name, dt * 1e3, 1.0 / dt, v), flush=True)
