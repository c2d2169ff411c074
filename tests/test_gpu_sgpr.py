"""GPU parity of SGPR (gpflow/models/sgpr.py) against the oracle, plus the reference's own relational
tests (tests/gpflow/models/test_sgpr.py:29-80)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as orc  # noqa: E402  (checker only)
from oracle import gp_oracle_grad as orcg  # noqa: E402  (checker only)


def _data(N, M, D, P, seed):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.2 * rng.normal(size=(N, P)); Z = rng.normal(size=(M, D))
    ls = 0.8 + 0.1 * np.arange(D)
    return X, Y, Z, dict(variance=1.3, lengthscales=ls, noise_variance=0.25)


def _model(X, Y, Z, kw, mean=None):
    import gpflow_amd as gpflow
    mf = gpflow.mean_functions.Constant(np.array([mean])) if mean is not None else None
    return gpflow.models.SGPR((X, Y), gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"]),
                              Z, noise_variance=kw["noise_variance"], mean_function=mf)


@pytest.mark.parametrize("N,M,D,P,mean", [(500, 60, 2, 1, None), (3000, 384, 4, 2, 0.3), (8192, 1024, 8, 1, None)])
def test_sgpr_elbo_and_predict_vs_oracle(gpu, N, M, D, P, mean):
    X, Y, Z, kw = _data(N, M, D, P, 1)
    m = _model(X, Y, Z, kw, mean)
    mo = 0.0 if mean is None else mean
    ref = orc.sgpr_elbo(X, Y, Z, mean=mo, **kw)
    got = float(m.elbo().cpu())
    assert abs(got - ref) <= 1e-8 * abs(ref), (got, ref)
    assert float(m.training_loss().cpu()) == -got or abs(float(m.training_loss().cpu()) + got) <= 1e-12 * abs(got)
    Xn = np.random.default_rng(2).normal(size=(257, D))
    fm, fv = m.predict_f(Xn)
    rm, rv = orc.sgpr_predict_f(X, Y, Z, Xn, mean=mo, **kw)
    np.testing.assert_allclose(fm.cpu().numpy(), rm, rtol=0, atol=1e-8 * max(1.0, np.abs(rm).max()))
    np.testing.assert_allclose(fv.cpu().numpy(), rv, rtol=0, atol=1e-8)
    if M <= 384:
        fm2, fc = m.predict_f(Xn[:100], full_cov=True)
        _, rc = orc.sgpr_predict_f(X, Y, Z, Xn[:100], mean=mo, full_cov=True, **kw)
        np.testing.assert_allclose(fc.cpu().numpy(), rc, rtol=0, atol=1e-8)
        with pytest.raises(NotImplementedError):
            m.predict_f(Xn, full_output_cov=True)
    if P == 1:
        ub = float(m.upper_bound().cpu())
        rub = orc.sgpr_upper_bound(X, Y, Z, mean=mo, **kw)
        assert abs(ub - rub) <= 1e-8 * abs(rub)
        assert got <= ub


def test_sgpr_qu_and_svgp_equivalence(gpu):
    """tests/gpflow/models/test_sgpr.py:29-80: q(u) from compute_qu == predict_f at Z; an un-whitened SVGP carrying that
    q(u) predicts like the SGPR."""
    import gpflow_amd as gpflow
    rng = np.random.RandomState(0)
    X = rng.randn(100, 2); Y = rng.randn(100, 1); Z = rng.randn(20, 2)
    kw = dict(variance=1.0, lengthscales=1.0, noise_variance=1.0)
    m = _model(X, Y, Z, kw)
    mu, cov = m.compute_qu()
    rmu, rcov = orc.sgpr_compute_qu(X, Y, Z, **kw)
    np.testing.assert_allclose(mu.cpu().numpy(), rmu, rtol=0, atol=1e-9)
    np.testing.assert_allclose(cov.cpu().numpy(), rcov, rtol=0, atol=1e-9)
    fz, fzc = m.predict_f(Z, full_cov=True)
    np.testing.assert_allclose(mu.cpu().numpy(), fz.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(cov.cpu().numpy()[None], fzc.cpu().numpy(), rtol=1e-4, atol=1e-4)
    q_sqrt = np.linalg.cholesky(cov.cpu().numpy())[None]
    svgp = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(), gpflow.likelihoods.Gaussian(1.0), Z,
                              q_mu=mu.cpu().numpy(), q_sqrt=q_sqrt, whiten=False)
    Xnew = np.random.RandomState(2).randn(100, 2)
    a, b = m.predict_f(Xnew), svgp.predict_f(Xnew)
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), atol=1e-4)
    np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), atol=1e-4)


def test_natural_gradient_step_and_svgp_vs_sgpr(gpu):
    """tests/gpflow/optimizers/test_natural_gradient.py:171 (test_svgp_vs_sgpr): ONE natural-gradient step of size 1 takes
    the SVGP bound (Gaussian likelihood) to the SGPR bound; a short step (0.1) equals the oracle's literal restatement."""
    import gpflow_amd as gpflow
    from oracle import gp_oracle_grad as orcg
    X, Y, Z, kw = _data(1200, 200, 3, 2, 5)
    rng = np.random.default_rng(6)
    q_mu = 0.3 * rng.normal(size=(200, 2))
    q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(200, 200))) + 0.6 * np.eye(200) for _ in range(2)])

    def svgp():
        return gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"]),
                                  gpflow.likelihoods.Gaussian(kw["noise_variance"]), Z.copy(), q_mu=q_mu.copy(),
                                  q_sqrt=q_sqrt.copy(), num_data=X.shape[0])
    sgpr = _model(X, Y, Z, kw)
    target = float(sgpr.elbo().cpu())
    m = svgp()
    before = float(m.elbo((X, Y)).cpu())
    assert abs(before - target) > 1.0
    gpflow.optimizers.NaturalGradient(1.0).minimize(m, (X, Y))
    after = float(m.elbo((X, Y)).cpu())
    assert abs(after - target) <= 1e-4, (before, after, target)
    # a short step against the oracle
    m2 = svgp()
    gpflow.optimizers.NaturalGradient(0.1).minimize(m2, (X, Y))
    _, g = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=X.shape[0], **kw)
    mu_r, sq_r = orcg.natgrad_step(q_mu, q_sqrt, -g["q_mu"], -g["q_sqrt"], 0.1)
    np.testing.assert_allclose(m2.q_mu.numpy(), mu_r, rtol=0, atol=1e-8 * max(1.0, np.abs(mu_r).max()))
    np.testing.assert_allclose(m2.q_sqrt.numpy(), sq_r, rtol=0, atol=1e-8)


def test_sgpr_gradients_and_scipy_fit(gpu):
    """SGPR.objective_and_grad vs the autograd oracle (chained to the unconstrained parameters), then the reference's
    test_sgpr_qu recipe (tests/gpflow/models/test_sgpr.py:29-44): optimise with Scipy, then q(u) == predict_f at Z."""
    import gpflow_amd as gpflow
    from oracle import gp_oracle_grad as orcg
    rng = np.random.RandomState(1)
    X = np.random.RandomState(0).randn(100, 2); Z = np.random.RandomState(0).randn(20, 2)
    Y = np.sin(X @ np.array([[-1.4], [0.5]])) + 0.5 * rng.randn(len(X), 1)
    m = gpflow.models.SGPR((X, Y), gpflow.kernels.SquaredExponential(), Z.copy())
    v, g = m.objective_and_grad()
    rv, rg = orcg.sgpr_elbo_value_and_grads(X, Y, Z, variance=1.0, lengthscales=1.0, noise_variance=1.0)
    assert abs(v - rv) <= 1e-9 * abs(rv)
    sp = gpflow.base.positive()
    u1 = sp.inverse(1.0)
    np.testing.assert_allclose(g[m.kernel.variance], rg["variance"] * sp.forward_grad(u1), rtol=1e-8)
    np.testing.assert_allclose(g[m.kernel.lengthscales], rg["lengthscales"].reshape(()) * sp.forward_grad(u1), rtol=1e-8)
    np.testing.assert_allclose(g[m.inducing_variable.Z], rg["Z"], rtol=0, atol=1e-8 * np.abs(rg["Z"]).max())
    tn = m.likelihood.variance.transform
    np.testing.assert_allclose(g[m.likelihood.variance], rg["noise_variance"] * tn.forward_grad(tn.inverse(1.0)), rtol=1e-8)
    res = gpflow.optimizers.Scipy().minimize(m, options=dict(maxiter=200))
    assert -res.fun > v + 5.0
    qu_mean, qu_cov = m.compute_qu()
    fz, fzc = m.predict_f(m.inducing_variable.Z.numpy(), full_cov=True)
    np.testing.assert_allclose(qu_mean.cpu().numpy(), fz.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(qu_cov.cpu().numpy().reshape(1, 20, 20), fzc.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_nextrows_golden_vectors(gpu):
    """The device paths of the rows built this round against the committed golden fixtures (tests/golden/
    nextrows_golden.npz: the reference's own SGPR / SVGP test fixtures; generator committed beside it)."""
    import os
    import gpflow_amd as gpflow
    from gpflow_amd import gradients, natgrad, ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "nextrows_golden.npz"))
    kw = dict(variance=1.0, lengthscales=1.0, noise_variance=1.0)
    m = _model(g["sgpr_X"], g["sgpr_Y"], g["sgpr_Z"], kw)
    np.testing.assert_allclose(float(m.elbo().cpu()), g["sgpr_elbo"], rtol=1e-9)
    np.testing.assert_allclose(float(m.upper_bound().cpu()), g["sgpr_upper"], rtol=1e-9)
    fm, fv = m.predict_f(g["sgpr_Xnew"])
    np.testing.assert_allclose(fm.cpu().numpy(), g["sgpr_mean"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(fv.cpu().numpy(), g["sgpr_var"], rtol=0, atol=1e-9)
    mu, cov = m.compute_qu()
    np.testing.assert_allclose(mu.cpu().numpy(), g["sgpr_qu_mean"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(cov.cpu().numpy(), g["sgpr_qu_cov"], rtol=0, atol=1e-9)
    _, gs, _ = gradients.sgpr_elbo_and_grad(ops.to_device(g["sgpr_Z"]), ops.to_device(g["sgpr_X"]), ops.to_device(g["sgpr_Y"]),
                                            jitter=1e-6, **kw)
    np.testing.assert_allclose(gs["Z"].cpu().numpy(), g["sgpr_g_Z"], rtol=0, atol=1e-8 * np.abs(g["sgpr_g_Z"]).max())
    np.testing.assert_allclose(gs["noise_variance"].cpu().numpy().reshape(()), g["sgpr_g_noise"], rtol=1e-8)
    t = ops.to_device
    F, gr, _ = gradients.svgp_elbo_and_grad(t(g["grad_Z"]), t(g["grad_X"]), t(g["grad_Y"]), t(g["grad_q_mu"]), t(g["grad_q_sqrt"]),
                                            jitter=1e-6, scale=200 / 20, **kw)
    np.testing.assert_allclose(float(F.cpu()[0]), g["grad_elbo"], rtol=1e-9)
    for k in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt"):
        ref = g[f"grad_g_{k}"]
        np.testing.assert_allclose(gr[k].cpu().numpy().reshape(ref.shape), ref, rtol=0, atol=1e-8 * max(1.0, np.abs(ref).max()))
    mu_n, sq_n = natgrad.natgrad_update(t(g["grad_q_mu"]), t(g["grad_q_sqrt"]), -gr["q_mu"], -gr["q_sqrt"], 0.3)
    np.testing.assert_allclose(mu_n.cpu().numpy(), g["nat_q_mu"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(sq_n.cpu().numpy(), g["nat_q_sqrt"], rtol=0, atol=1e-8)


def test_trainer_natgrad_hybrid(gpu):
    """SVGPTrainer(natgrad_gamma=1) with frozen hyper-parameters: one step takes the device-resident q(u) to the optimum
    (next ELBO == SGPR bound); with trainable hyper-parameters 25 hybrid steps beat 25 plain Adam steps."""
    import gpflow_amd as gpflow
    from gpflow_amd import training
    X, Y, Z, kw = _data(1500, 128, 3, 1, 9)

    def svgp():
        return gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"]),
                                  gpflow.likelihoods.Gaussian(kw["noise_variance"]), Z.copy(), num_data=X.shape[0])
    m = svgp()
    for p in (m.kernel.variance, m.kernel.lengthscales, m.likelihood.variance, m.inducing_variable.Z):
        gpflow.set_trainable(p, False) if hasattr(p, "_walk") else setattr(p, "_trainable", False)
    tr = training.SVGPTrainer(m, natgrad_gamma=1.0)
    tr.step((X, Y))
    f1 = float(tr.step((X, Y)).cpu()[0])
    target = float(_model(X, Y, Z, kw).elbo().cpu())
    assert abs(f1 - target) <= 1e-4, (f1, target)
    a, b = training.SVGPTrainer(svgp(), learning_rate=1e-2, natgrad_gamma=0.5), training.SVGPTrainer(svgp(), learning_rate=1e-2)
    for _ in range(25):
        fa, fb = a.step((X, Y)), b.step((X, Y))
    assert float(fa.cpu()[0]) > float(fb.cpu()[0])


@pytest.mark.parametrize("M,P", [(96, 3), (300, 2)])
def test_natgrad_batched_over_latents_and_xi_sqrt_mean_var(gpu, M, P):
    """natgrad.natgrad_update runs all P latents as batched launches (one gpk_potrf(batch = P) per factorisation, batched
    triangular-K GEMMs; optimizers/natgrad.py:429-516 is batched [P, M, M] too) for both xi transforms -- XiNat (:98-136) and
    XiSqrtMeanVar (:139-173) -- against the literal restatement with torch autograd (reverse mode through
    expectation_to_meanvarsqrt, forward mode through natural_to_meanvarsqrt); and through gpflow.optimizers.NaturalGradient."""
    from gpflow_amd import natgrad, ops
    rng = np.random.default_rng(M + P)
    q_mu = rng.normal(size=(M, P))
    q_sqrt = np.tril(0.1 * rng.normal(size=(P, M, M))) + np.eye(M) * (0.6 + 0.3 * rng.random(size=(P, 1, 1)))
    g_mu = rng.normal(size=(M, P))
    g_sqrt = np.tril(0.05 * rng.normal(size=(P, M, M)))   # (small enough that the XiNat precision stays positive definite)
    t = ops.to_device
    for xi in ("XiNat", "XiSqrtMeanVar"):
        for gamma in (0.05, 0.3):
            mu_n, sq_n = natgrad.natgrad_update(t(q_mu), t(q_sqrt), t(g_mu), t(g_sqrt), gamma, xi_transform=xi)
            mu_r, sq_r = orcg.natgrad_step(q_mu, q_sqrt, g_mu, g_sqrt, gamma, xi_transform=xi)
            np.testing.assert_allclose(mu_n.cpu().numpy(), mu_r, rtol=0, atol=1e-8 * max(1.0, np.abs(mu_r).max()), err_msg=xi)
            np.testing.assert_allclose(sq_n.cpu().numpy(), sq_r, rtol=0, atol=1e-8 * max(1.0, np.abs(sq_r).max()), err_msg=xi)
            assert np.all(np.triu(sq_n.cpu().numpy(), 1) == 0)
