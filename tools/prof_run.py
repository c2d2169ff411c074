"""Workload for rocprofv3 --kernel-trace --stats: a few SVGP steps (Cm), one N=16384 GPR LML, or (c5sep) a few ELBO
evaluations of BASELINE config C5 with SeparateIndependent kernels through the model surface, or (train) a few forward +
reverse evaluations of the SVGP step at Cm."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
rng = np.random.default_rng(0)
dev = ops.device()
if which in ("svgp", "both"):
    m, B, d = 2048, 8192, 8
    Z = ops.to_device(rng.normal(size=(m, d)))
    Xb = ops.to_device(rng.normal(size=(B, d)))
    Yb = ops.to_device(rng.normal(size=(B, 1)))
    q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
    q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    ws = ops.svgp_elbo_workspace(m, B, d, 1, False)
    for _ in range(3):
        out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls,
                                        noise_variance=0.1, jitter=1e-6, ws=ws)
    torch.cuda.synchronize()
    print("svgp", out.cpu().numpy())
if which in ("gpr", "both"):
    n, d = 16384, 8
    X = ops.to_device(rng.normal(size=(n, d)))
    Y = ops.to_device(rng.normal(size=(n, 1)))
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    for _ in range(2):
        out, info = ops.gpr_lml(X, Y, variance=1.0, lengthscales=ls, noise_variance=0.1)
    torch.cuda.synchronize()
    print("gpr", out.cpu().numpy(), info.cpu().numpy())
if which == "c5sep":
    import gpflow_amd as gpflow
    m, b, d, p = 1024, 8192, 8, 4
    Zh = rng.normal(size=(m, d))
    X = ops.to_device(rng.normal(size=(b, d)))
    Y = ops.to_device(rng.normal(size=(b, p)))
    qm = 0.1 * rng.normal(size=(m, p))
    qs = np.stack([np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m) for _ in range(p)])
    ksep = gpflow.kernels.SeparateIndependent([gpflow.kernels.SquaredExponential(variance=v, lengthscales=l)
                                               for v, l in zip([1.0, 0.8, 1.2, 0.9], [2.4, 2.8, 3.2, 3.6])])
    iv = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Zh))
    mp = gpflow.models.SVGP(ksep, gpflow.likelihoods.Gaussian(0.1), iv, q_mu=qm, q_sqrt=qs, num_latent_gps=p, num_data=1_000_000)
    for _ in range(4):
        v = float(mp.elbo((X, Y)))
    print("c5sep", v)
if which == "train":
    from gpflow_amd import gradients
    m, B, d = 2048, 8192, 8
    Z = ops.to_device(rng.normal(size=(m, d)))
    Xb = ops.to_device(rng.normal(size=(B, d)))
    Yb = ops.to_device(np.sin(rng.normal(size=(B, 1))))
    q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
    q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    for _ in range(4):
        out = gradients.svgp_elbo_and_grad(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6,
                                           scale=100.0)
    torch.cuda.synchronize()
    print("train", float(out[0]))
if which == "unwh":
    import gpflow_amd as gpflow
    m, B, d = 2048, 8192, 8
    Zh = rng.normal(size=(m, d))
    X = ops.to_device(rng.normal(size=(B, d)))
    Y = ops.to_device(np.sin(rng.normal(size=(B, 1))))
    qm = 0.1 * rng.normal(size=(m, 1))
    qs = (np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None]
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    mu = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=ls), gpflow.likelihoods.Gaussian(0.1), Zh,
                            q_mu=qm, q_sqrt=qs, whiten=False, num_data=1_000_000)
    for _ in range(4):
        v = float(mu.elbo((X, Y)))
    print("unwh", v)
