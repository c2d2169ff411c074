"""Generates tests/golden/nextrows_golden.npz FROM THE ORACLES (oracle/gp_oracle.py, oracle/gp_oracle_grad.py) for the
SURVEY 8f rows built this round, on the fixtures of the reference's own tests (the reference cannot be imported here:
no TensorFlow).  Re-run:  python tests/golden/make_golden_next.py
Fixtures:
  sgpr_*  tests/gpflow/models/test_sgpr.py:22-35   RandomState(0): X[100,2], Z[20,2]; Y = sin(X [-1.4, 0.5]^T) + 0.5 RandomState(1)
  grad_*  tests/gpflow/models/test_svgp.py:28-36 style RandomState(0): X[20,1], Y[20,2]^2, Z[3,1], q_mu, q_sqrt (as hotpath_golden)
  nat_*   one XiNat step (gamma = 0.3) on the grad_* fixture (optimizers/natgrad.py:280-368)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as orc  # noqa: E402
from oracle import gp_oracle_grad as orcg  # noqa: E402

out = {}
X = np.random.RandomState(0).randn(100, 2); Z = np.random.RandomState(0).randn(20, 2)
rng = np.random.RandomState(1)
Y = np.sin(X @ np.array([[-1.4], [0.5]])) + 0.5 * rng.randn(len(X), 1)
Xnew = np.random.RandomState(2).randn(30, 2)
kw = dict(variance=1.0, lengthscales=1.0, noise_variance=1.0)
fm, fv = orc.sgpr_predict_f(X, Y, Z, Xnew, **kw)
mu, cov = orc.sgpr_compute_qu(X, Y, Z, **kw)
v, g = orcg.sgpr_elbo_value_and_grads(X, Y, Z, **kw)
out.update(sgpr_X=X, sgpr_Y=Y, sgpr_Z=Z, sgpr_Xnew=Xnew, sgpr_elbo=orc.sgpr_elbo(X, Y, Z, **kw),
           sgpr_upper=orc.sgpr_upper_bound(X, Y, Z, **kw), sgpr_mean=fm, sgpr_var=fv, sgpr_qu_mean=mu, sgpr_qu_cov=cov,
           sgpr_g_variance=g["variance"], sgpr_g_lengthscales=g["lengthscales"], sgpr_g_noise=g["noise_variance"], sgpr_g_Z=g["Z"])

rng = np.random.RandomState(0)
X = rng.randn(20, 1); Y = rng.randn(20, 2) ** 2; Z = rng.randn(3, 1)
q_mu = rng.randn(3, 2); q_sqrt = np.array([np.tril(rng.randn(3, 3)) for _ in range(2)])
q_sqrt[:, np.arange(3), np.arange(3)] = np.abs(q_sqrt[:, np.arange(3), np.arange(3)]) + 0.1
v, g = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=200, **kw)
out.update(grad_X=X, grad_Y=Y, grad_Z=Z, grad_q_mu=q_mu, grad_q_sqrt=q_sqrt, grad_elbo=v,
           **{f"grad_g_{k}": val for k, val in g.items()})
mu_n, sq_n = orcg.natgrad_step(q_mu, q_sqrt, -g["q_mu"], -g["q_sqrt"], 0.3)
out.update(nat_q_mu=mu_n, nat_q_sqrt=sq_n)
v, g = orcg.gpr_lml_value_and_grads(X, Y[:, :1], **kw)
out.update(gprgrad_lml=v, **{f"gprgrad_g_{k}": val for k, val in g.items()})

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nextrows_golden.npz")
np.savez(path, **out)
print("wrote", path, sorted(out))
