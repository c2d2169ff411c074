"""Generates tests/golden/hotpath_golden.npz FROM THE ORACLE (oracle/gp_oracle.py) on the fixtures the
reference's own tests use (the reference stores no golden values for this path and cannot be imported
here: no TensorFlow).  Re-run:  python tests/golden/make_golden.py
Fixtures:
  gpr_*   tests/gpflow/models/test_gpr.py:21-30 style   RandomState(0), N=10, D=1, ls=2.0, var=1.0
  svgp_*  tests/gpflow/models/test_svgp.py:28-36 style  RandomState(0), X[20,1], Y[20,2]^2, Z[3,1]
  kl_*    tests/gpflow/test_kullback_leiblers.py:106-118 RandomState(0), M=5
  c1_*    default_rng(1), N=512, D=2, RBF(1,1), noise 0.1
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as orc  # noqa: E402

out = {}
rng = np.random.RandomState(0)
X = rng.randn(10, 1); Y = np.sin(X) + 0.1 * rng.randn(10, 1); Xnew = rng.randn(7, 1)
kw = dict(variance=1.0, lengthscales=2.0, noise_variance=1.0)
out.update(gpr_X=X, gpr_Y=Y, gpr_Xnew=Xnew, gpr_var=1.0, gpr_ls=2.0, gpr_noise=1.0,
           gpr_lml=orc.gpr_log_marginal_likelihood(X, Y, **kw))
mu, var = orc.gpr_predict_f(X, Y, Xnew, **kw)
out.update(gpr_mu=mu, gpr_var_pred=var)

rng = np.random.RandomState(0)
X = rng.randn(20, 1); Y = rng.randn(20, 2) ** 2; Z = rng.randn(3, 1)
q_mu = rng.randn(3, 2); q_sqrt = np.array([np.tril(rng.randn(3, 3)) for _ in range(2)])
q_sqrt[:, np.arange(3), np.arange(3)] = np.abs(q_sqrt[:, np.arange(3), np.arange(3)]) + 0.1
out.update(svgp_X=X, svgp_Y=Y, svgp_Z=Z, svgp_q_mu=q_mu, svgp_q_sqrt=q_sqrt, svgp_var=1.0, svgp_ls=1.0, svgp_noise=1.0)
for w in (0, 1):
    out[f"svgp_elbo_w{w}"] = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, variance=1.0, lengthscales=1.0,
                                            noise_variance=1.0, whiten=bool(w))

rng = np.random.RandomState(0)
M = 5
mu = rng.randn(M, 4); sqrt = np.array([np.tril(rng.randn(M, M)) for _ in range(4)])
A = rng.randn(M, M); K = A @ A.T + 1e-6 * np.eye(M)
out.update(kl_mu=mu, kl_sqrt=sqrt, kl_K=K, kl_white=orc.gauss_kl(mu, sqrt), kl_K_val=orc.gauss_kl(mu, sqrt, K))

rng = np.random.default_rng(1)
X = rng.normal(size=(512, 2)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(512, 1))
out.update(c1_X=X, c1_Y=Y, c1_lml=orc.gpr_log_marginal_likelihood(X, Y, variance=1.0, lengthscales=1.0, noise_variance=0.1))

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_golden.npz")
np.savez(path, **out)
print("wrote", path, {k: (np.asarray(v).shape if np.ndim(v) else float(v)) for k, v in out.items() if k.endswith(("lml", "w0", "w1", "white", "val"))})
