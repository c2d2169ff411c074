"""Regenerates the golden fixtures of tests/golden/ FROM THE REAL REFERENCE (GPflow + TensorFlow) -- the pin this build
could not apply itself: TensorFlow / TFP / check_shapes are not installable in the build image, so
hotpath_golden.npz was generated from oracle/gp_oracle.py (make_golden.py) and oracle-vs-reference parity is only
relational.  On any machine with `pip install gpflow` (2.9.x, TF >= 2.4):

    python tests/golden/make_golden_tf.py            # writes tests/golden/hotpath_golden_tf.npz
    python tests/golden/make_golden_tf.py --check    # additionally compares with the committed oracle-made fixture

Same inputs (RandomState(0) / default_rng(1) fixtures of the reference's own tests, listed in make_golden.py), same keys;
every value is computed by the reference's public API:
    gpflow.models.GPR.log_marginal_likelihood / predict_f        (gpflow/models/gpr.py:91-107, 178-190)
    gpflow.models.SVGP.elbo, whiten False / True                 (gpflow/models/svgp.py:166-181)
    gpflow.kullback_leiblers.gauss_kl with K = None and K given  (gpflow/kullback_leiblers.py:59-165)
`--check` asserts 1e-9 relative agreement on every scalar and 1e-9 absolute on the predictions: with that run green,
the oracle (and through it the HIP path) is pinned to the reference numerically, not just structurally.
"""
import os
import sys

import numpy as np


def main(check: bool) -> None:
    import gpflow  # noqa: F401  (the real one)
    import tensorflow as tf  # noqa: F401
    from gpflow.kullback_leiblers import gauss_kl

    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    rng = np.random.RandomState(0)
    X = rng.randn(10, 1); Y = np.sin(X) + 0.1 * rng.randn(10, 1); Xnew = rng.randn(7, 1)
    m = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=2.0), noise_variance=1.0)
    mu, var = m.predict_f(Xnew)
    out.update(gpr_X=X, gpr_Y=Y, gpr_Xnew=Xnew, gpr_var=1.0, gpr_ls=2.0, gpr_noise=1.0,
               gpr_lml=float(m.log_marginal_likelihood()), gpr_mu=mu.numpy(), gpr_var_pred=var.numpy())

    rng = np.random.RandomState(0)
    X = rng.randn(20, 1); Y = rng.randn(20, 2) ** 2; Z = rng.randn(3, 1)
    q_mu = rng.randn(3, 2); q_sqrt = np.array([np.tril(rng.randn(3, 3)) for _ in range(2)])
    q_sqrt[:, np.arange(3), np.arange(3)] = np.abs(q_sqrt[:, np.arange(3), np.arange(3)]) + 0.1
    out.update(svgp_X=X, svgp_Y=Y, svgp_Z=Z, svgp_q_mu=q_mu, svgp_q_sqrt=q_sqrt, svgp_var=1.0, svgp_ls=1.0, svgp_noise=1.0)
    for w in (0, 1):
        s = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=1.0),
                               gpflow.likelihoods.Gaussian(variance=1.0), Z.copy(), q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(),
                               whiten=bool(w), num_latent_gps=2)
        out[f"svgp_elbo_w{w}"] = float(s.elbo((X, Y)))

    rng = np.random.RandomState(0)
    M = 5
    mu = rng.randn(M, 4); sqrt = np.array([np.tril(rng.randn(M, M)) for _ in range(4)])
    A = rng.randn(M, M); K = A @ A.T + 1e-6 * np.eye(M)
    out.update(kl_mu=mu, kl_sqrt=sqrt, kl_K=K, kl_white=float(gauss_kl(mu, sqrt)), kl_K_val=float(gauss_kl(mu, sqrt, K)))

    rng = np.random.default_rng(1)
    X = rng.normal(size=(512, 2)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(512, 1))
    m1 = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(), noise_variance=0.1)
    out.update(c1_X=X, c1_Y=Y, c1_lml=float(m1.log_marginal_likelihood()))

    path = os.path.join(here, "hotpath_golden_tf.npz")
    np.savez(path, **out)
    print("wrote", path)
    if check:
        ref = np.load(os.path.join(here, "hotpath_golden.npz"))
        for k in ("gpr_lml", "svgp_elbo_w0", "svgp_elbo_w1", "kl_white", "kl_K_val", "c1_lml"):
            a, b = float(out[k]), float(ref[k])
            assert abs(a - b) <= 1e-9 * abs(b), (k, a, b)
            print(f"  {k}: reference {a:.12f}  oracle {b:.12f}  rel {abs(a - b) / abs(b):.2e}")
        for k in ("gpr_mu", "gpr_var_pred"):
            d = float(np.abs(np.asarray(out[k]) - ref[k]).max())
            assert d <= 1e-9, (k, d)
            print(f"  {k}: max abs diff {d:.2e}")
        print("oracle-made fixtures agree with GPflow + TensorFlow: parity pinned")


if __name__ == "__main__":
    try:
        main("--check" in sys.argv)
    except ImportError as e:
        raise SystemExit(f"this script needs the real reference (pip install gpflow tensorflow): {e}")
