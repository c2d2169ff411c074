"""CPU checks for the gradient path:
  * the autograd oracle (oracle/gp_oracle_grad.py) is pinned to the NumPy oracle: same value, gradients == central
    finite differences of gp_oracle.svgp_elbo;
  * the hand-written adjoint in gpflow_amd/gradients.py -- a composition of device primitives -- is validated with the
    primitives swapped for their CPU emulation (tests/fake_ops.py), which reproduces what each HIP kernel reads and
    writes (triangular K ranges, skipped upper tiles).  The GPU tests run the same function on the real kernels.
"""
import numpy as np
import pytest

from oracle import gp_oracle as orc
from oracle import gp_oracle_grad as orcg


def _problem(M, B, D, P, seed, ard=True):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(B, D))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(B, P))
    Z = rng.normal(size=(M, D))
    q_mu = 0.3 * rng.normal(size=(M, P))
    q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(M, M))) + 0.6 * np.eye(M) for _ in range(P)])
    ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D)) if ard else 1.3
    return X, Y, Z, q_mu, q_sqrt, dict(variance=1.3, lengthscales=ls, noise_variance=0.2)


def test_autograd_oracle_value_matches_numpy_oracle():
    X, Y, Z, q_mu, q_sqrt, kw = _problem(40, 90, 3, 2, 0)
    v, _ = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=1000, **kw)
    ref = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, whiten=True, num_data=1000, **kw)
    assert abs(v - ref) <= 1e-12 * abs(ref)


def test_autograd_oracle_gradients_match_finite_differences():
    X, Y, Z, q_mu, q_sqrt, kw = _problem(12, 30, 2, 2, 1)
    _, g = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=200, **kw)

    def f(**over):
        a = dict(X=X, Y=Y, Z=Z, q_mu=q_mu, q_sqrt=q_sqrt, **kw)
        a.update(over)
        return orc.svgp_elbo(a["X"], a["Y"], a["Z"], a["q_mu"], a["q_sqrt"], variance=a["variance"],
                             lengthscales=a["lengthscales"], noise_variance=a["noise_variance"], whiten=True, num_data=200)

    def fd(name, idx=None, h=1e-6):
        base = np.array(dict(Z=Z, q_mu=q_mu, q_sqrt=q_sqrt, **kw)[name], dtype=np.float64)
        def at(delta):
            v = base.copy()
            if idx is None:
                v = v + delta
            else:
                v[idx] += delta
            return f(**{name: v if v.ndim else float(v)})
        return (at(h) - at(-h)) / (2 * h)

    for name, idx in [("variance", None), ("noise_variance", None), ("lengthscales", (0,)), ("lengthscales", (1,)),
                      ("Z", (3, 1)), ("Z", (0, 0)), ("q_mu", (5, 1)), ("q_sqrt", (0, 4, 2)), ("q_sqrt", (1, 7, 7))]:
        got = g[name] if idx is None else g[name][idx]
        got = float(np.asarray(got).reshape(-1)[0]) if idx is None else float(got)
        ref = fd(name, idx)
        assert abs(got - ref) <= 2e-6 * max(1.0, abs(ref)), (name, idx, got, ref)
    # upper triangle of q_sqrt is ignored (band_part, conditionals/util.py:151): zero gradient
    assert g["q_sqrt"][0, 2, 5] == 0.0


@pytest.mark.parametrize("M,B,D,P,ard", [(150, 300, 3, 2, True), (260, 140, 2, 1, False), (64, 500, 4, 3, True)])
def test_adjoint_composition_on_emulated_primitives(monkeypatch, M, B, D, P, ard):
    import torch
    from gpflow_amd import gradients
    import fake_ops
    monkeypatch.setattr(gradients, "ops", fake_ops)
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, 2, ard)
    # junk above the diagonal of q_sqrt must be ignored (band_part)
    q_in = q_sqrt + np.triu(np.ones((M, M)), 1)[None] * 0.37
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    scale = 1000.0 / B
    F, g, info = gradients.svgp_elbo_and_grad(t(Z), t(X), t(Y), t(q_mu), t(q_in), jitter=1e-6, scale=scale,
                                              mean_const=0.1, **kw)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=1000, mean=0.1, **kw)
    assert abs(float(F[0]) - v) <= 1e-10 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, ref = g[name].numpy(), np.asarray(go[name])
        tol = 1e-8 * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(got.reshape(ref.shape) if got.size == ref.size else got, ref, rtol=0, atol=tol,
                                   err_msg=name)


def _small_model(M=40, B=120, D=2, P=2, seed=5):
    import gpflow_amd as gpflow
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, seed)
    k = gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"])
    m = gpflow.models.SVGP(k, gpflow.likelihoods.Gaussian(kw["noise_variance"]), Z.copy(), q_mu=q_mu.copy(),
                           q_sqrt=q_sqrt.copy(), num_data=2000)
    return m, X, Y


def _patch_ops(monkeypatch):
    import fake_ops
    from gpflow_amd import gradients, training
    monkeypatch.setattr(gradients, "ops", fake_ops)
    monkeypatch.setattr(training, "ops", fake_ops)


def test_trainer_first_step_is_tf_adam_on_oracle_gradients(monkeypatch):
    """One SVGPTrainer step == the tf.keras Adam update rule applied to the autograd-oracle gradient, chained through the
    softplus transforms for the positive parameters (emulated primitives; the GPU test repeats it on the kernels)."""
    from gpflow_amd import training
    _patch_ops(monkeypatch)
    m, X, Y = _small_model()
    u0 = {n: np.array(p.unconstrained_variable, copy=True) for n, p in
          dict(variance=m.kernel.variance, lengthscales=m.kernel.lengthscales, noise=m.likelihood.variance).items()}
    Z0, qm0, qs0 = m.inducing_variable.Z.numpy().copy(), m.q_mu.numpy().copy(), m.q_sqrt.numpy().copy()
    v, g = orcg.svgp_elbo_value_and_grads(X, Y, Z0, qm0, qs0, variance=m.kernel.variance.numpy(),
                                          lengthscales=m.kernel.lengthscales.numpy(),
                                          noise_variance=m.likelihood.variance.numpy(), num_data=2000)
    tr = training.SVGPTrainer(m, learning_rate=1e-2)
    F = tr.step((X, Y))
    assert abs(float(F[0]) - v) <= 1e-10 * abs(v)
    tr.sync_to_model()
    lr_t = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)

    def adam1(p, grad_loss):     # first step from zero moments
        mm, vv = 0.1 * grad_loss, 0.001 * grad_loss ** 2
        return p - lr_t * mm / (np.sqrt(vv) + 1e-7)

    np.testing.assert_allclose(m.q_mu.numpy(), adam1(qm0, -g["q_mu"]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(m.q_sqrt.numpy(), np.tril(adam1(qs0, -g["q_sqrt"])), rtol=0, atol=1e-9)
    np.testing.assert_allclose(m.inducing_variable.Z.numpy(), adam1(Z0, -g["Z"]), rtol=0, atol=1e-9)
    for name, par, gc in [("variance", m.kernel.variance, g["variance"]), ("lengthscales", m.kernel.lengthscales, g["lengthscales"]),
                          ("noise", m.likelihood.variance, g["noise_variance"])]:
        gu = -np.asarray(gc).reshape(u0[name].shape) * par.transform.forward_grad(u0[name])
        np.testing.assert_allclose(par.unconstrained_variable, adam1(u0[name], gu), rtol=0, atol=1e-9, err_msg=name)


def test_trainer_improves_the_elbo_and_respects_trainable(monkeypatch):
    from gpflow_amd import training
    _patch_ops(monkeypatch)
    m, X, Y = _small_model(M=20, B=80, P=1)
    m.inducing_variable.Z._trainable = False
    Z0 = m.inducing_variable.Z.numpy().copy()
    tr = training.SVGPTrainer(m, learning_rate=5e-2)
    vals = [float(tr.step((X, Y))[0]) for _ in range(40)]
    assert vals[-1] > vals[0] + 10.0, (vals[0], vals[-1])
    tr.sync_to_model()
    np.testing.assert_array_equal(m.inducing_variable.Z.numpy(), Z0)
    assert np.all(np.triu(m.q_sqrt.numpy()[0], 1) == 0.0)


def test_gpr_autograd_oracle_matches_numpy_oracle_and_fd():
    rng = np.random.default_rng(3)
    X = rng.normal(size=(40, 2)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(40, 2))
    kw = dict(variance=1.4, lengthscales=np.array([0.8, 1.2]), noise_variance=0.15)
    v, g = orcg.gpr_lml_value_and_grads(X, Y, mean=0.2, **kw)
    ref = orc.gpr_log_marginal_likelihood(X, Y, mean=0.2, **kw)
    assert abs(v - float(np.sum(ref))) <= 1e-12 * abs(v)
    h = 1e-6
    for name in ("variance", "noise_variance"):
        up, dn = dict(kw), dict(kw)
        up[name] += h; dn[name] -= h
        fd = (np.sum(orc.gpr_log_marginal_likelihood(X, Y, mean=0.2, **up))
              - np.sum(orc.gpr_log_marginal_likelihood(X, Y, mean=0.2, **dn))) / (2 * h)
        assert abs(float(g[name]) - fd) <= 2e-6 * max(1.0, abs(fd)), name


@pytest.mark.parametrize("N,D,P,ard", [(200, 3, 2, True), (130, 2, 1, False)])
def test_gpr_adjoint_composition_on_emulated_primitives(monkeypatch, N, D, P, ard):
    import torch
    from gpflow_amd import gradients
    import fake_ops
    monkeypatch.setattr(gradients, "ops", fake_ops)
    rng = np.random.default_rng(4)
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P))
    ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D)) if ard else 1.1
    kw = dict(variance=1.4, lengthscales=ls, noise_variance=0.15)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    F, g, info = gradients.gpr_lml_and_grad(t(X), t(Y), mean_const=0.2, **kw)
    v, go = orcg.gpr_lml_value_and_grads(X, Y, mean=0.2, **kw)
    assert abs(float(F[0]) - v) <= 1e-10 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "mean_const"):
        got, ref = g[name].numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=1e-8 * max(1.0, np.abs(ref).max()), err_msg=name)


def test_scipy_optimizer_fits_gpr_on_emulated_primitives(monkeypatch):
    """gpflow.optimizers.Scipy().minimize on a GPR: packs the unconstrained trainables, L-BFGS-B with the device gradient
    (emulated primitives here).  The optimum must match scipy run directly on the autograd oracle."""
    import scipy.optimize
    import fake_ops
    import gpflow_amd as gpflow
    from gpflow_amd import gradients
    from gpflow_amd.models import gpr as gpr_mod
    monkeypatch.setattr(gradients, "ops", fake_ops)
    monkeypatch.setattr(gpr_mod, "ops", fake_ops)
    rng = np.random.default_rng(6)
    X = rng.uniform(-2, 2, size=(60, 1)); Y = np.sin(2 * X) + 0.1 * rng.normal(size=(60, 1))
    m = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(), noise_variance=0.5)
    v0, g = m.log_marginal_likelihood_and_grad()
    assert set(g) == {m.kernel.variance, m.kernel.lengthscales, m.likelihood.variance}
    res = gpflow.optimizers.Scipy().minimize(m, options=dict(maxiter=200))
    assert res.success and -res.fun > v0 + 10

    sp = gpflow.base.positive()
    def f(u):   # the same objective from the oracle, in unconstrained space (all three use the default positive transform)
        var, ls, nv = sp.forward(u[0]), sp.forward(u[1]), m.likelihood.variance.transform.forward(u[2])
        v, go = orcg.gpr_lml_value_and_grads(X, Y, variance=var, lengthscales=ls, noise_variance=nv)
        gu = np.array([go["variance"].item() * sp.forward_grad(u[0]), go["lengthscales"].item() * sp.forward_grad(u[1]),
                       go["noise_variance"].item() * m.likelihood.variance.transform.forward_grad(u[2])])
        return -v, -gu
    ref = scipy.optimize.minimize(f, np.array([sp.inverse(1.0), sp.inverse(1.0), m.likelihood.variance.transform.inverse(0.5)]),
                                  jac=True, method="L-BFGS-B", options=dict(maxiter=200))
    assert abs(res.fun - ref.fun) <= 1e-6 * abs(ref.fun)
    np.testing.assert_allclose(m.kernel.lengthscales.numpy(), sp.forward(ref.x[1]), rtol=1e-4)


def test_sgpr_statistics_composition_on_emulated_primitives(monkeypatch):
    """SGPR: shard statistics -> sum -> replicated tail, against the oracle; two row shards summed by
    hand stand in for the all-reduce (the gloo test runs the real collective)."""
    import torch
    import fake_ops
    from gpflow_amd import gradients
    from gpflow_amd.models import sgpr
    monkeypatch.setattr(gradients, "ops", fake_ops)
    monkeypatch.setattr(sgpr, "ops", fake_ops)
    rng = np.random.default_rng(8)
    N, M, D, P = 700, 150, 3, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P)); Z = rng.normal(size=(M, D))
    ls = np.array([0.9, 1.1, 1.3])
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    kw = dict(variance=1.2, lengthscales=ls, family="SquaredExponential", jitter=1e-6, mean_const=0.3)
    packed = None
    for lo, hi in [(0, 401), (401, 700)]:
        L, invd, pk = sgpr.shard_statistics(t(Z), t(X[lo:hi]), t(Y[lo:hi]), **kw)
        packed = pk if packed is None else packed + pk
    elbo = sgpr.elbo_from_statistics(packed, M, P, N, variance=1.2, noise_variance=0.25)
    ref = orc.sgpr_elbo(X, Y, Z, variance=1.2, lengthscales=ls, noise_variance=0.25, mean=0.3)
    assert abs(float(elbo) - ref) <= 1e-10 * abs(ref)
    # single output: the upper bound
    L, invd, pk1 = sgpr.shard_statistics(t(Z), t(X), t(Y[:, :1]), **kw)
    ub = sgpr.upper_bound_from_statistics(pk1, M, N, variance=1.2, noise_variance=0.25)
    ref_ub = orc.sgpr_upper_bound(X, Y[:, :1], Z, variance=1.2, lengthscales=ls, noise_variance=0.25, mean=0.3)
    assert abs(float(ub) - ref_ub) <= 1e-10 * abs(ref_ub)


def test_sgpr_heteroskedastic_statistics_and_upper_bound_on_emulated_primitives(monkeypatch):
    """One noise variance per data row (Gaussian(scale = Function), sgpr.py:207-211 and :108-145): statistics of two row shards summed
    by hand, ELBO and upper bound against the oracle -- the bound needs a SECOND pass with noise_rows = sigma_n^2 + c, where c comes
    from the summed first pass."""
    import torch
    import fake_ops
    from gpflow_amd import gradients
    from gpflow_amd.models import sgpr
    monkeypatch.setattr(gradients, "ops", fake_ops)
    monkeypatch.setattr(sgpr, "ops", fake_ops)
    rng = np.random.default_rng(81)
    N, M, D = 500, 60, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, 1)); Z = rng.normal(size=(M, D))
    nv = (0.2 + 0.1 * np.abs(X[:, 0]) + 0.05 * X[:, 1] ** 2)            # sigma_n^2 at the data inputs
    ls = np.array([0.9, 1.2])
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    kw = dict(variance=1.1, lengthscales=ls, family="SquaredExponential", jitter=1e-6, mean_const=0.2)
    shards = [(0, 263), (263, 500)]
    packed = sum(sgpr.shard_statistics(t(Z), t(X[lo:hi]), t(Y[lo:hi]), noise_rows=t(nv[lo:hi]), **kw)[2] for lo, hi in shards)
    okw = dict(variance=1.1, lengthscales=ls, noise_variance=nv, mean=0.2)
    elbo = sgpr.elbo_from_statistics(packed, M, 1, N, variance=1.1, noise_variance=None)
    ref = orc.sgpr_elbo(X, Y, Z, **okw)
    assert abs(float(elbo) - ref) <= 1e-10 * abs(ref)
    c_tr = N * 1.1 - float(packed[M * M + M + 4])
    packed_cn = sum(sgpr.shard_statistics(t(Z), t(X[lo:hi]), t(Y[lo:hi]), noise_rows=t(nv[lo:hi] + c_tr), **kw)[2] for lo, hi in shards)
    ub = sgpr.upper_bound_heteroskedastic(packed, packed_cn, M, N)
    ref_ub = orc.sgpr_upper_bound(X, Y, Z, **okw)
    assert abs(float(ub) - ref_ub) <= 1e-10 * abs(ref_ub)
    assert ref < ref_ub
    # a constant noise variance given per row reproduces the constant-noise bound
    L, invd, pk1 = sgpr.shard_statistics(t(Z), t(X), t(Y), **kw)
    ub_c = sgpr.upper_bound_from_statistics(pk1, M, N, variance=1.1, noise_variance=0.3)
    pkh = sgpr.shard_statistics(t(Z), t(X), t(Y), noise_rows=t(np.full(N, 0.3)), **kw)[2]
    c0 = N * 1.1 - float(pkh[M * M + M + 4])
    pkc = sgpr.shard_statistics(t(Z), t(X), t(Y), noise_rows=t(np.full(N, 0.3 + c0)), **kw)[2]
    assert abs(float(sgpr.upper_bound_heteroskedastic(pkh, pkc, M, N)) - float(ub_c)) <= 1e-10 * abs(float(ub_c))


@pytest.mark.parametrize("op", ["add", "mul"])
def test_kernel_combination_with_diagonal_q_sqrt_on_emulated_primitives(monkeypatch, op):
    """Sum / Product of stationary kernels (kernels/base.py:216-220, 305-315) under an SVGP with q_diag = True (svgp.py:90-148),
    whitened and un-whitened: value and every gradient of the hand-written reverse pass against autograd over the restated
    model.  (The covariance spec and the q_diag branches of the reverse pass do not interact; this pins that.)"""
    import torch
    import fake_ops
    from gpflow_amd import gradients
    monkeypatch.setattr(gradients, "ops", fake_ops)
    rng = np.random.default_rng(32)
    N, D, M, P = 120, 3, 30, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P))
    Z = X[:M] + 0.05 * rng.normal(size=(M, D))
    q_mu = 0.2 * rng.normal(size=(M, P)); q_sqrt = 0.4 + 0.2 * np.abs(rng.normal(size=(M, P)))
    members = [("SquaredExponential", 1.2, np.array([0.9, 1.1, 1.3])), ("Matern32", 0.7, np.array(0.8))]
    cols = [[0, 1, 2], [1]] if op == "mul" else None      # (the Product: its second member over one input column)
    mem = members if cols is None else [members[0], ("Matern32", 0.7, np.array(0.8))]
    spec = gradients.KernelSpec(mem, op, cols) if cols is not None else gradients.KernelSpec(mem, op)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    for name, fn in (("svgp", gradients.svgp_elbo_and_grad), ("svgp_unwhitened", gradients.svgp_elbo_and_grad_unwhitened)):
        F, g, info = fn(t(Z), t(X), t(Y), t(q_mu), t(q_sqrt), noise_variance=0.2, jitter=1e-6, scale=5.0, kernel_spec=spec)
        rv, rg = orcg.combination_value_and_grads(name, X, Y, mem, op, noise_variance=0.2, Z=Z, q_mu=q_mu, q_sqrt=q_sqrt,
                                                  num_data=5 * N, cols=cols)
        assert int(info[0]) == 0 and abs(float(F[0]) - rv) <= 1e-10 * abs(rv)
        for k in ("variance", "noise_variance", "Z", "q_mu", "q_sqrt"):
            ref = np.asarray(rg[k])
            np.testing.assert_allclose(np.asarray(g[k]).reshape(ref.shape), ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=k)
        for i in range(2):
            ref = np.asarray(rg["lengthscales"][i])
            np.testing.assert_allclose(g["lengthscales"][i].numpy().reshape(ref.shape), ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))


def test_natgrad_update_on_emulated_primitives_and_svgp_vs_sgpr(monkeypatch):
    """(i) the written-out natural-gradient step == the literal restatement of natgrad.py with autograd through the
    parameter conversions; (ii) tests/gpflow/optimizers/test_natural_gradient.py:171 (test_svgp_vs_sgpr): with a Gaussian
    likelihood ONE step of size 1 takes the SVGP bound to the SGPR bound."""
    import torch
    import fake_ops
    from gpflow_amd import gradients, natgrad
    monkeypatch.setattr(gradients, "ops", fake_ops)
    monkeypatch.setattr(natgrad, "ops", fake_ops)
    X, Y, Z, q_mu, q_sqrt, kw = _problem(140, 400, 2, 2, 9)
    N = X.shape[0]
    v, g = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=N, **kw)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    for gamma in (0.1, 1.0):
        mu_n, sq_n = natgrad.natgrad_update(t(q_mu), t(q_sqrt), t(-g["q_mu"]), t(-g["q_sqrt"]), gamma)
        mu_r, sq_r = orcg.natgrad_step(q_mu, q_sqrt, -g["q_mu"], -g["q_sqrt"], gamma)
        np.testing.assert_allclose(mu_n.numpy(), mu_r, rtol=0, atol=1e-9 * max(1.0, np.abs(mu_r).max()))
        np.testing.assert_allclose(sq_n.numpy(), sq_r, rtol=0, atol=1e-9)
        # XiSqrtMeanVar (natgrad.py:139-173): the step in (q_mu, q_sqrt) itself, = S g_mu and L Phi(L^T g_L) written out,
        # against forward-mode autograd through the restated natural_to_meanvarsqrt
        mu_x, sq_x = natgrad.natgrad_update(t(q_mu), t(q_sqrt), t(-g["q_mu"]), t(-g["q_sqrt"]), gamma, xi_transform="XiSqrtMeanVar")
        mu_xr, sq_xr = orcg.natgrad_step(q_mu, q_sqrt, -g["q_mu"], -g["q_sqrt"], gamma, xi_transform="XiSqrtMeanVar")
        np.testing.assert_allclose(mu_x.numpy(), mu_xr, rtol=0, atol=1e-9 * max(1.0, np.abs(mu_xr).max()))
        np.testing.assert_allclose(sq_x.numpy(), sq_xr, rtol=0, atol=1e-9)
    with pytest.raises(NotImplementedError):
        natgrad.natgrad_update(t(q_mu), t(q_sqrt), t(-g["q_mu"]), t(-g["q_sqrt"]), 0.1, xi_transform="XiSomethingElse")
    after = orc.svgp_elbo(X, Y, Z, mu_n.numpy(), sq_n.numpy(), whiten=True, num_data=N, **kw)
    sgpr = orc.sgpr_elbo(X, Y, Z, **kw)
    assert abs(v - sgpr) > 1.0                   # different before
    assert abs(after - sgpr) <= 1e-4             # equal after one step of size 1


@pytest.mark.parametrize("N,M,D,P,ard", [(400, 150, 3, 2, True), (300, 64, 2, 1, False)])
def test_sgpr_adjoint_composition_on_emulated_primitives(monkeypatch, N, M, D, P, ard):
    import torch
    import fake_ops
    from gpflow_amd import gradients
    monkeypatch.setattr(gradients, "ops", fake_ops)
    rng = np.random.default_rng(12)
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P)); Z = rng.normal(size=(M, D))
    ls = 0.8 + 0.1 * np.arange(D) if ard else 1.1
    kw = dict(variance=1.2, lengthscales=ls, noise_variance=0.3)
    v, go = orcg.sgpr_elbo_value_and_grads(X, Y, Z, mean=0.2, **kw)
    assert abs(v - orc.sgpr_elbo(X, Y, Z, mean=0.2, **kw)) <= 1e-11 * abs(v)      # the autograd oracle is the NumPy oracle
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    F, g, info = gradients.sgpr_elbo_and_grad(t(Z), t(X), t(Y), jitter=1e-6, mean_const=0.2, **kw)
    assert abs(float(F[0]) - v) <= 1e-10 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "mean_const"):
        got, ref = g[name].numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=1e-8 * max(1.0, np.abs(ref).max()), err_msg=name)


def test_trainer_natgrad_hybrid_reaches_the_sgpr_bound_in_one_step(monkeypatch):
    """SVGPTrainer(natgrad_gamma=1): (q_mu, q_sqrt) by a natural-gradient step, the rest by Adam.  With the hyper-parameters
    and Z frozen, one step puts q(u) at the optimum: the next ELBO evaluation equals the SGPR bound (oracle)."""
    import fake_ops
    from gpflow_amd import natgrad, training
    _patch_ops(monkeypatch)
    monkeypatch.setattr(natgrad, "ops", fake_ops)
    m, X, Y = _small_model(M=30, B=150, P=1, seed=15)
    for p in (m.kernel.variance, m.kernel.lengthscales, m.likelihood.variance, m.inducing_variable.Z):
        p._trainable = False
    m.num_data = X.shape[0]
    tr = training.SVGPTrainer(m, natgrad_gamma=1.0)
    f0 = float(tr.step((X, Y))[0])
    f1 = float(tr.step((X, Y))[0])
    ref = orc.sgpr_elbo(X, Y, m.inducing_variable.Z.numpy(), variance=float(m.kernel.variance.numpy()),
                        lengthscales=m.kernel.lengthscales.numpy(), noise_variance=float(m.likelihood.variance.numpy()))
    assert abs(f0 - ref) > 1.0 and abs(f1 - ref) <= 1e-4, (f0, f1, ref)


@pytest.mark.parametrize("M,B,D,P,ard", [(150, 300, 3, 2, True), (64, 200, 2, 1, False)])
def test_unwhitened_adjoint_on_emulated_primitives(monkeypatch, M, B, D, P, ard):
    """whiten=False: autograd oracle == NumPy oracle (value), then the hand-written adjoint on the emulated primitives."""
    import torch
    import fake_ops
    from gpflow_amd import gradients
    monkeypatch.setattr(gradients, "ops", fake_ops)
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, 17, ard)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=1000, mean=0.1, whiten=False, **kw)
    ref = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, whiten=False, num_data=1000, mean=0.1, **kw)
    assert abs(v - ref) <= 1e-10 * abs(ref)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    F, g, info = gradients.svgp_elbo_and_grad_unwhitened(t(Z), t(X), t(Y), t(q_mu), t(q_sqrt), jitter=1e-6, scale=1000.0 / B,
                                                         mean_const=0.1, **kw)
    assert abs(float(F[0]) - v) <= 1e-9 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, ref_g = g[name].numpy(), np.asarray(go[name])
        tol = 1e-7 * max(1.0, np.abs(ref_g).max())
        np.testing.assert_allclose(got.reshape(ref_g.shape), ref_g, rtol=0, atol=tol, err_msg=name)


# ----------------------------------------------------------------------------- Matern families (stationaries.py:254-313)
MATERN_TOL = {"Matern12": 2e-6, "Matern32": 1e-8, "Matern52": 1e-8}
# (Matern12: d exp(-r)/dr2 = -exp(-r)/(2r) is unbounded at r -> 0.  The autograd oracle differentiates the expansion
#  formula, whose diagonal r2_ii is rounding noise of either sign: where it comes out positive the oracle picks up
#  noise * 1/r ~ 1e-8 relative; the product writes the diagonal's factor as exact zeros (r2_ii = 0 identically).)


@pytest.mark.parametrize("family", ["Matern12", "Matern32", "Matern52"])
def test_matern_autograd_oracle_value_and_finite_differences(family):
    X, Y, Z, q_mu, q_sqrt, kw = _problem(12, 30, 2, 2, 21)
    v, g = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=200, family=family, **kw)
    ref = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, whiten=True, num_data=200, kernel=family, **kw)
    assert abs(v - ref) <= 1e-12 * abs(ref)

    def f(**over):
        a = dict(Z=Z, **kw)
        a.update(over)
        return orc.svgp_elbo(X, Y, a["Z"], q_mu, q_sqrt, variance=a["variance"], lengthscales=a["lengthscales"],
                             noise_variance=a["noise_variance"], whiten=True, num_data=200, kernel=family)

    # Matern12 as the reference writes it is not smooth at rounding level: K_ii = variance * exp(-sqrt(max(r2_ii, 1e-36)))
    # with r2_ii = rounding noise of the expansion formula, i.e. the VALUE jitters by ~1e-8 per diagonal entry whenever an
    # input moves.  Finite differences need a step far above that (and get a correspondingly loose tolerance).
    h, tol = (1e-3, 2e-3) if family == "Matern12" else (1e-6, 5e-6)
    for name, idx in [("variance", None), ("lengthscales", (0,)), ("lengthscales", (1,)), ("Z", (3, 1)), ("Z", (0, 0))]:
        base = np.array(dict(Z=Z, **kw)[name], dtype=np.float64)
        def at(delta):
            w = base.copy()
            if idx is None:
                w = w + delta
            else:
                w[idx] += delta
            return f(**{name: w if w.ndim else float(w)})
        fd = (at(h) - at(-h)) / (2 * h)
        got = float(np.asarray(g[name]).reshape(-1)[0]) if idx is None else float(g[name][idx])
        assert abs(got - fd) <= tol * max(1.0, abs(fd)), (family, name, idx, got, fd)


@pytest.mark.parametrize("family", ["Matern12", "Matern32", "Matern52"])
@pytest.mark.parametrize("whiten", [True, False])
def test_matern_svgp_adjoint_on_emulated_primitives(monkeypatch, family, whiten):
    import torch
    from gpflow_amd import gradients
    import fake_ops
    monkeypatch.setattr(gradients, "ops", fake_ops)
    M, B, D, P = 90, 260, 3, 2
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, 22, True)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    fn = gradients.svgp_elbo_and_grad if whiten else gradients.svgp_elbo_and_grad_unwhitened
    F, g, info = fn(t(Z), t(X), t(Y), t(q_mu), t(q_sqrt), jitter=1e-6, scale=1000.0 / B, mean_const=0.1, family=family, **kw)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=1000, mean=0.1, whiten=whiten, family=family, **kw)
    assert int(info) == 0 and abs(float(F[0]) - v) <= 1e-10 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, ref = g[name].numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=MATERN_TOL[family] * max(1.0, np.abs(ref).max()),
                                   err_msg=f"{family} {name}")


@pytest.mark.parametrize("family", ["Matern12", "Matern32", "Matern52"])
def test_matern_gpr_and_sgpr_adjoints_on_emulated_primitives(monkeypatch, family):
    import torch
    from gpflow_amd import gradients
    import fake_ops
    monkeypatch.setattr(gradients, "ops", fake_ops)
    rng = np.random.default_rng(23)
    N, M, D, P = 220, 60, 3, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P)); Z = rng.normal(size=(M, D))
    kw = dict(variance=1.4, lengthscales=np.sqrt(D) * (0.8 + 0.05 * np.arange(D)), noise_variance=0.15)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    F, g, info = gradients.gpr_lml_and_grad(t(X), t(Y), mean_const=0.2, family=family, **kw)
    v, go = orcg.gpr_lml_value_and_grads(X, Y, mean=0.2, family=family, **kw)
    assert abs(v - float(np.sum(orc.gpr_log_marginal_likelihood(X, Y, mean=0.2, kernel=family, **kw)))) <= 1e-11 * abs(v)
    assert abs(float(F[0]) - v) <= 1e-10 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "mean_const"):
        got, ref = g[name].numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=MATERN_TOL[family] * max(1.0, np.abs(ref).max()),
                                   err_msg=f"gpr {family} {name}")
    F, g, info = gradients.sgpr_elbo_and_grad(t(Z), t(X), t(Y), jitter=1e-6, mean_const=0.2, family=family, **kw)
    v, go = orcg.sgpr_elbo_value_and_grads(X, Y, Z, mean=0.2, family=family, **kw)
    assert abs(float(F[0]) - v) <= 1e-10 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "mean_const"):
        got, ref = g[name].numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=MATERN_TOL[family] * max(1.0, np.abs(ref).max()),
                                   err_msg=f"sgpr {family} {name}")


@pytest.mark.parametrize("family", ["SquaredExponential", "Matern52"])
def test_unwhitened_q_diag_adjoint_on_emulated_primitives(monkeypatch, family):
    """whiten=False with q_diag=True (q_sqrt [M, P] standard deviations): the autograd oracle's forward equals the NumPy
    oracle (gauss_kl with K and a diagonal q: kullback_leiblers.py:131-152), and the hand-written adjoint equals autograd."""
    import torch
    from gpflow_amd import gradients
    import fake_ops
    monkeypatch.setattr(gradients, "ops", fake_ops)
    M, B, D, P = 80, 230, 3, 2
    X, Y, Z, q_mu, _, kw = _problem(M, B, D, P, 24, True)
    q = 0.3 + np.abs(np.random.default_rng(25).normal(size=(M, P)))
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q, num_data=1000, mean=0.1, whiten=False, family=family, **kw)
    ref = orc.svgp_elbo(X, Y, Z, q_mu, q, num_data=1000, mean=0.1, whiten=False, kernel=family, **kw)
    assert abs(v - ref) <= 1e-9 * abs(ref)     # (Kuu^-1 at jitter 1e-6: two LAPACK routes through cond ~1e6)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
    F, g, info = gradients.svgp_elbo_and_grad_unwhitened(t(Z), t(X), t(Y), t(q_mu), t(q), jitter=1e-6, scale=1000.0 / B,
                                                         mean_const=0.1, family=family, **kw)
    assert int(info) == 0 and abs(float(F[0]) - v) <= 1e-9 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, refg = g[name].numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(refg.shape), refg, rtol=0, atol=1e-8 * max(1.0, np.abs(refg).max()), err_msg=name)
