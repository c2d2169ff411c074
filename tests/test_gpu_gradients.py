"""GPU parity of the hand-written reverse pass (gpflow_amd/gradients.py) against the autograd oracle.
Tolerance: 1e-8 of the largest entry of each gradient (fp64; the same bar as the forward value)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle_grad as orcg  # noqa: E402  (checker only)


def _problem(M, B, D, P, seed, ard=True):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(B, D))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(B, P))
    Z = rng.normal(size=(M, D))
    q_mu = 0.3 * rng.normal(size=(M, P))
    q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(M, M))) + 0.6 * np.eye(M) for _ in range(P)])
    ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D)) if ard else 1.3
    return X, Y, Z, q_mu, q_sqrt, dict(variance=1.3, lengthscales=ls, noise_variance=0.2)


def test_kernel_matrix_hadamard(gpu):
    from gpflow_amd import ops
    rng = np.random.default_rng(0)
    for n1, n2, d in [(70, 130, 3), (256, 512, 8), (1, 5, 1)]:
        A, Bm, G = rng.normal(size=(n1, d)), rng.normal(size=(n2, d)), rng.normal(size=(n1, n2))
        ls = 0.7 + 0.1 * np.arange(d)
        K = ops.kernel_matrix(ops.to_device(A), ops.to_device(Bm), variance=1.7, lengthscales=ls)
        H = ops.kernel_matrix_hadamard(ops.to_device(A), ops.to_device(Bm), ops.to_device(G), variance=1.7, lengthscales=ls)
        np.testing.assert_array_equal(H.cpu().numpy(), K.cpu().numpy() * G)   # same K bits, one rounding each


@pytest.mark.parametrize("n,chunks", [(256, 4), (384, 2), (2048, 4)])
def test_cholesky_adjoint_k_split_products(gpu, monkeypatch, n, chunks):
    """gradients.cholesky_adjoint with its three triangular x triangular products split along K (ops.gemm_nt k_split, the path
    M >= 1024 takes) against the unsplit products and against torch autograd through torch.linalg.cholesky."""
    import torch
    from gpflow_amd import gradients, ops
    if n > 512 and not torch.cuda.is_available():
        pytest.skip("full size on the GPU only")
    rng = np.random.default_rng(5)
    R = rng.normal(size=(n, n))
    K = R @ R.T / n + np.eye(n)
    L = np.linalg.cholesky(K)
    Lbar = np.tril(rng.normal(size=(n, n)))
    t = ops.to_device
    LT, LinvT, Lb = t(L.T.copy()), t(np.linalg.inv(L).T.copy()), t(Lbar)
    monkeypatch.setattr(gradients, "TRI_PRODUCT_MIN_N", 64)
    monkeypatch.setattr(gradients, "TRI_PRODUCT_CHUNKS", chunks)
    got = gradients.cholesky_adjoint(LT, LinvT, Lb).cpu().numpy()
    monkeypatch.setattr(gradients, "TRI_PRODUCT_CHUNKS", 1)
    unsplit = gradients.cholesky_adjoint(LT, LinvT, Lb).cpu().numpy()
    Kt = torch.tensor(K, dtype=torch.float64, requires_grad=True)
    (torch.linalg.cholesky(Kt) * torch.tensor(Lbar)).sum().backward()
    ref = 0.5 * (Kt.grad + Kt.grad.T).numpy()
    tol = 1e-9 * max(1.0, np.abs(ref).max())
    np.testing.assert_allclose(got, unsplit, rtol=0, atol=tol)
    np.testing.assert_allclose(got, ref, rtol=0, atol=tol)


@pytest.mark.parametrize("M,B,D,P,ard", [(150, 300, 3, 2, True), (260, 140, 2, 1, False), (64, 500, 4, 3, True),
                                         (640, 2048, 8, 1, True)])
def test_svgp_elbo_and_grad_vs_autograd_oracle(gpu, M, B, D, P, ard):
    from gpflow_amd import gradients, ops
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, 2, ard)
    q_in = q_sqrt + np.triu(np.ones((M, M)), 1)[None] * 0.37   # junk above the diagonal is ignored (band_part)
    t = ops.to_device
    F, g, info = gradients.svgp_elbo_and_grad(t(Z), t(X), t(Y), t(q_mu), t(q_in), jitter=1e-6, scale=1000.0 / B,
                                              mean_const=0.1, **kw)
    ops.check_info(info)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=1000, mean=0.1, **kw)
    assert abs(float(F.cpu()[0]) - v) <= 1e-9 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, ref = g[name].cpu().numpy(), np.asarray(go[name])
        tol = 1e-8 * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=tol, err_msg=name)


@pytest.mark.parametrize("M,B,D,P,ard", [(150, 300, 3, 2, True), (64, 200, 2, 1, False), (384, 1024, 8, 1, True)])
def test_unwhitened_svgp_elbo_and_grad_vs_autograd_oracle(gpu, M, B, D, P, ard):
    """whiten=False (conditionals/util.py:137-139, kullback_leiblers.py:98-165 with K = Kuu): value and every gradient
    of the hand-written adjoint against the autograd oracle; tolerance 1e-7 of the largest entry (the un-whitened
    chain goes through the explicit Lm^-1 twice, one more kappa(Lm) than the whitened path)."""
    from gpflow_amd import gradients, ops
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, 17, ard)
    t = ops.to_device
    F, g, info = gradients.svgp_elbo_and_grad_unwhitened(t(Z), t(X), t(Y), t(q_mu), t(q_sqrt), jitter=1e-6,
                                                         scale=1000.0 / B, mean_const=0.1, **kw)
    ops.check_info(info)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=1000, mean=0.1, whiten=False, **kw)
    assert abs(float(F.reshape(-1).cpu()[0]) - v) <= 1e-9 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, ref = g[name].cpu().numpy(), np.asarray(go[name])
        tol = 1e-7 * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=tol, err_msg=name)


def test_unwhitened_model_gradients_and_trainer(gpu):
    """SVGP(whiten=False).elbo_and_grad reaches the un-whitened adjoint (value == the model's own composed forward,
    gradients vs the autograd oracle); SVGPTrainer steps it, trains a Constant mean, and refuses a failed step."""
    import gpflow_amd as gpflow
    from gpflow_amd import training
    from gpflow_amd._lib import GpkError
    X, Y, Z, q_mu, q_sqrt, kw = _problem(48, 256, 2, 1, 23)
    k = gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"])
    m = gpflow.models.SVGP(k, gpflow.likelihoods.Gaussian(kw["noise_variance"]), Z.copy(), q_mu=q_mu.copy(),
                           q_sqrt=q_sqrt.copy(), num_data=2000, whiten=False,
                           mean_function=gpflow.mean_functions.Constant(0.25))
    v, g = m.elbo_and_grad((X, Y))
    rv, rg = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=2000, mean=0.25, whiten=False, **kw)
    assert abs(v - rv) <= 1e-9 * abs(rv)
    assert abs(v - float(m.elbo((X, Y)).cpu())) <= 1e-9 * abs(v)
    np.testing.assert_allclose(g[m.q_mu], rg["q_mu"], rtol=0, atol=1e-7 * np.abs(rg["q_mu"]).max())
    np.testing.assert_allclose(g[m.inducing_variable.Z], rg["Z"], rtol=0, atol=1e-7 * np.abs(rg["Z"]).max())
    assert m.mean_function.c in g                      # Constant.c is trainable by default
    np.testing.assert_allclose(np.ravel(g[m.mean_function.c]), np.ravel(rg["mean_const"]), rtol=1e-7)
    tr = training.SVGPTrainer(m, learning_rate=2e-2)
    vals = [float(tr.step((X, Y)).cpu()[0]) for _ in range(12)]
    assert abs(vals[0] - rv) <= 1e-9 * abs(rv) and vals[-1] > vals[0]
    assert tr.last_info == 0
    tr.sync_to_model()
    assert abs(float(np.ravel(m.mean_function.c.numpy())[0]) - 0.25) > 1e-3   # the mean constant was trained
    # a step whose Kuu is not positive definite must raise and leave the variables untouched
    m2 = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1e13, lengthscales=1.0),
                            gpflow.likelihoods.Gaussian(0.1), np.zeros((40, 2)), num_data=2000)
    tr2 = training.SVGPTrainer(m2)
    before = tr2.dev["q_mu"].clone()
    with pytest.raises(GpkError):
        tr2.step((X, Y))
    assert tr2.last_info != 0 and tr2.opt.t == 0
    np.testing.assert_array_equal(tr2.dev["q_mu"].cpu().numpy(), before.cpu().numpy())


def _small_model(M, B, D, P, seed):
    import gpflow_amd as gpflow
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, seed)
    k = gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"])
    m = gpflow.models.SVGP(k, gpflow.likelihoods.Gaussian(kw["noise_variance"]), Z.copy(), q_mu=q_mu.copy(),
                           q_sqrt=q_sqrt.copy(), num_data=5000)
    return m, X, Y


def test_trainer_first_step_is_tf_adam_on_oracle_gradients(gpu):
    from gpflow_amd import training
    m, X, Y = _small_model(150, 400, 3, 2, 7)
    hp = dict(variance=m.kernel.variance, lengthscales=m.kernel.lengthscales, noise=m.likelihood.variance)
    u0 = {n: np.array(p.unconstrained_variable, copy=True) for n, p in hp.items()}
    Z0, qm0, qs0 = m.inducing_variable.Z.numpy().copy(), m.q_mu.numpy().copy(), m.q_sqrt.numpy().copy()
    v, g = orcg.svgp_elbo_value_and_grads(X, Y, Z0, qm0, qs0, variance=m.kernel.variance.numpy(),
                                          lengthscales=m.kernel.lengthscales.numpy(),
                                          noise_variance=m.likelihood.variance.numpy(), num_data=5000)
    tr = training.SVGPTrainer(m, learning_rate=1e-2)
    F = tr.step((X, Y))
    assert abs(float(F.cpu()[0]) - v) <= 1e-9 * abs(v)
    tr.sync_to_model()
    lr_t = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)

    def adam1(p, grad_loss):     # first step from zero moments (tf.keras Adam, epsilon 1e-7)
        return p - lr_t * (0.1 * grad_loss) / (np.sqrt(0.001 * grad_loss ** 2) + 1e-7)

    # the update is lr * sign-like (|m|/sqrt(v) = 1 at step 1) wherever |g| >> 1e-7: compare where that holds
    for got, p0, gr in [(m.q_mu.numpy(), qm0, -g["q_mu"]), (m.inducing_variable.Z.numpy(), Z0, -g["Z"])]:
        np.testing.assert_allclose(got, adam1(p0, gr), rtol=0, atol=1e-7)
    low = np.tril(np.ones_like(qs0[0])) > 0
    np.testing.assert_allclose(m.q_sqrt.numpy()[:, low], adam1(qs0, -g["q_sqrt"])[:, low], rtol=0, atol=1e-7)
    for name, par, gc in [("variance", hp["variance"], g["variance"]), ("lengthscales", hp["lengthscales"], g["lengthscales"]),
                          ("noise", hp["noise"], g["noise_variance"])]:
        gu = -np.asarray(gc).reshape(u0[name].shape) * par.transform.forward_grad(u0[name])
        np.testing.assert_allclose(par.unconstrained_variable, adam1(u0[name], gu), rtol=0, atol=1e-7, err_msg=name)


def test_trainer_improves_elbo_and_agrees_with_fused_forward(gpu):
    """40 Adam steps raise the ELBO; after sync the model's own (fused C-ABI) ELBO equals the value the gradient path
    computes at the same parameters -- two independent device implementations of the forward pass."""
    from gpflow_amd import training
    m, X, Y = _small_model(256, 1500, 4, 1, 8)
    e0 = float(m.elbo((X, Y)).cpu())
    tr = training.SVGPTrainer(m, learning_rate=2e-2)
    f_first = float(tr.step((X, Y)).cpu()[0])
    assert abs(f_first - e0) <= 1e-9 * abs(e0)
    for _ in range(39):
        tr.step((X, Y))
    tr.sync_to_model()
    e1 = float(m.elbo((X, Y)).cpu())
    f_next = float(tr.step((X, Y)).cpu()[0])
    assert e1 > e0 + 1.0, (e0, e1)
    assert abs(f_next - e1) <= 1e-9 * abs(e1), (f_next, e1)


@pytest.mark.parametrize("N,D,P,ard", [(200, 3, 2, True), (700, 2, 1, False), (1100, 4, 1, True), (4224, 8, 1, True)])
def test_gpr_lml_and_grad_vs_autograd_oracle(gpu, N, D, P, ard):
    """N = 4224 also covers the large-matrix schedule of the factorisation (768-column outer panels, CU-masked bulk
    stream) TOGETHER with many extra rows (the identity rows that return L^-T)."""
    from gpflow_amd import gradients, ops
    rng = np.random.default_rng(4)
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P))
    ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D)) if ard else 1.1
    kw = dict(variance=1.4, lengthscales=ls, noise_variance=0.15)
    F, g, info = gradients.gpr_lml_and_grad(ops.to_device(X), ops.to_device(Y), mean_const=0.2, **kw)
    ops.check_info(info)
    v, go = orcg.gpr_lml_value_and_grads(X, Y, mean=0.2, **kw)
    assert abs(float(F.cpu()[0]) - v) <= 1e-9 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "mean_const"):
        got, ref = g[name].cpu().numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=1e-8 * max(1.0, np.abs(ref).max()), err_msg=name)


def test_scipy_optimizer_fits_gpr(gpu):
    """gpflow.optimizers.Scipy().minimize(GPR) with the device gradient reaches the optimum scipy finds on the oracle."""
    import scipy.optimize
    import gpflow_amd as gpflow
    rng = np.random.default_rng(6)
    X = rng.uniform(-2, 2, size=(300, 2)); Y = np.sin(2 * X[:, :1]) * np.cos(X[:, 1:]) + 0.1 * rng.normal(size=(300, 1))
    m = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(lengthscales=[1.0, 1.0]), noise_variance=0.5)
    v0 = float(m.log_marginal_likelihood().cpu())
    res = gpflow.optimizers.Scipy().minimize(m, options=dict(maxiter=300))
    assert -res.fun > v0 + 10
    assert abs(float(m.log_marginal_likelihood().cpu()) + res.fun) <= 1e-9 * abs(res.fun)   # fused forward agrees

    sp, tn = gpflow.base.positive(), m.likelihood.variance.transform

    def f(u):
        var, ls, nv = sp.forward(u[0]), sp.forward(u[1:3]), tn.forward(u[3])
        v, go = orcg.gpr_lml_value_and_grads(X, Y, variance=var, lengthscales=ls, noise_variance=nv)
        gu = np.concatenate([[go["variance"].item() * sp.forward_grad(u[0])], go["lengthscales"] * sp.forward_grad(u[1:3]),
                             [go["noise_variance"].item() * tn.forward_grad(u[3])]])
        return -v, -gu
    x0 = np.concatenate([[sp.inverse(1.0)], sp.inverse(np.ones(2)), [tn.inverse(0.5)]])
    ref = scipy.optimize.minimize(f, x0, jac=True, method="L-BFGS-B", options=dict(maxiter=300))
    assert abs(res.fun - ref.fun) <= 1e-6 * abs(ref.fun), (res.fun, ref.fun)
    np.testing.assert_allclose(m.kernel.lengthscales.numpy(), sp.forward(ref.x[1:3]), rtol=1e-3)


def test_svgp_elbo_and_grad_chained_to_unconstrained_and_scipy(gpu):
    """SVGP.elbo_and_grad: gradients in the UNCONSTRAINED space of every trainable parameter (softplus for the positive
    ones, fill-triangular for q_sqrt) against finite differences of the model's own fused ELBO; then Scipy on a fixed
    batch raises the ELBO."""
    import gpflow_amd as gpflow
    m, X, Y = _small_model(40, 300, 2, 1, 11)
    v, g = m.elbo_and_grad((X, Y))
    assert abs(v - float(m.elbo((X, Y)).cpu())) <= 1e-9 * abs(v)
    assert set(g) == {m.kernel.variance, m.kernel.lengthscales, m.likelihood.variance, m.inducing_variable.Z, m.q_mu, m.q_sqrt}
    h = 1e-5
    for par, idx in [(m.kernel.variance, ()), (m.kernel.lengthscales, (1,)), (m.likelihood.variance, ()),
                     (m.q_sqrt, (0, 17)), (m.q_sqrt, (0, 400)), (m.q_mu, (3, 0)), (m.inducing_variable.Z, (5, 1))]:
        u0 = np.array(par.unconstrained_variable, dtype=np.float64, copy=True)
        vals = []
        for d in (h, -h):
            u = u0.copy()
            u[idx] += d
            par.assign_unconstrained(u)
            vals.append(float(m.elbo((X, Y)).cpu()))
        par.assign_unconstrained(u0)
        fd = (vals[0] - vals[1]) / (2 * h)
        got = float(np.asarray(g[par])[idx])
        assert abs(got - fd) <= 1e-5 * max(1.0, abs(fd)), (par.name, idx, got, fd)
    res = gpflow.optimizers.Scipy().minimize(m, (X, Y), options=dict(maxiter=30))
    assert -res.fun > v + 10.0


def test_shared_independent_svgp_gradients_and_trainer(gpu):
    """BASELINE config C5's structure (SharedIndependent kernel + SharedIndependentInducingVariables, P latents sharing
    Kuu / Kuf): elbo_and_grad against the autograd oracle, agreement with the fused forward, and the trainer."""
    import gpflow_amd as gpflow
    from gpflow_amd import training
    X, Y, Z, q_mu, q_sqrt, kw = _problem(60, 300, 2, 3, 21)
    k = gpflow.kernels.SharedIndependent(gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"]),
                                         output_dim=3)
    iv = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Z.copy()))
    m = gpflow.models.SVGP(k, gpflow.likelihoods.Gaussian(kw["noise_variance"]), iv, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(),
                           num_data=3000)
    v, g = m.elbo_and_grad((X, Y))
    rv, rg = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=3000, **kw)
    assert abs(v - rv) <= 1e-9 * abs(rv)
    assert abs(v - float(m.elbo((X, Y)).cpu())) <= 1e-9 * abs(v)
    np.testing.assert_allclose(g[m.q_mu], rg["q_mu"], rtol=0, atol=1e-8 * np.abs(rg["q_mu"]).max())
    np.testing.assert_allclose(g[iv.inducing_variable.Z], rg["Z"], rtol=0, atol=1e-8 * np.abs(rg["Z"]).max())
    tr = training.SVGPTrainer(m, learning_rate=2e-2)
    vals = [float(tr.step((X, Y)).cpu()[0]) for _ in range(15)]
    assert abs(vals[0] - rv) <= 1e-9 * abs(rv) and vals[-1] > vals[0]


def test_active_dims_and_separate_independent_model_gradients(gpu):
    """SVGP.elbo_and_grad beyond the plain kernel: (i) `active_dims` (the kernel sees a column subset; dF/dZ is zero
    elsewhere, kernels/base.py:90-109), (ii) SeparateIndependent kernels over shared and over separate inducing points
    (one single-output problem per latent, conditionals/util.py:566-629) -- against the autograd oracle on the
    equivalent sliced / per-latent problems, and against the model's own fused forward."""
    import gpflow_amd as gpflow
    X, Y, Z, q_mu, q_sqrt, kw = _problem(50, 240, 4, 1, 31)
    dims = [3, 1]
    ls = np.array([0.9, 1.4])
    k = gpflow.kernels.SquaredExponential(variance=1.3, lengthscales=ls, active_dims=dims)
    m = gpflow.models.SVGP(k, gpflow.likelihoods.Gaussian(0.2), Z.copy(), q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), num_data=2000)
    v, g = m.elbo_and_grad((X, Y))
    rv, rg = orcg.svgp_elbo_value_and_grads(X[:, dims], Y, Z[:, dims], q_mu, q_sqrt, num_data=2000, variance=1.3, lengthscales=ls,
                                            noise_variance=0.2)
    assert abs(v - rv) <= 1e-9 * abs(rv) and abs(v - float(m.elbo((X, Y)).cpu())) <= 1e-9 * abs(v)
    gz = np.asarray(g[m.inducing_variable.Z])
    np.testing.assert_allclose(gz[:, dims], rg["Z"], rtol=0, atol=1e-8 * np.abs(rg["Z"]).max())
    assert np.all(gz[:, [0, 2]] == 0.0)
    np.testing.assert_allclose(g[m.q_mu], rg["q_mu"], rtol=0, atol=1e-8 * np.abs(rg["q_mu"]).max())
    # (ii) three latents, one kernel each; shared Z, then separate Z
    P = 3
    X, Y, Z, q_mu, q_sqrt, kw = _problem(40, 200, 2, P, 33)
    vs, lss = [1.0, 0.7, 1.4], [np.array([0.8, 1.1]), np.array([1.3, 0.9]), np.array([1.0, 1.6])]
    for separate_z in (False, True):
        Zs = [Z + 0.1 * p for p in range(P)] if separate_z else [Z] * P
        kern = gpflow.kernels.SeparateIndependent([gpflow.kernels.SquaredExponential(variance=vs[p], lengthscales=lss[p]) for p in range(P)])
        if separate_z:
            iv = gpflow.inducing_variables.SeparateIndependentInducingVariables([gpflow.inducing_variables.InducingPoints(z.copy()) for z in Zs])
        else:
            iv = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Z.copy()))
        m = gpflow.models.SVGP(kern, gpflow.likelihoods.Gaussian(0.2), iv, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), num_latent_gps=P,
                               num_data=2000)
        v, g = m.elbo_and_grad((X, Y))
        rv, gq, gnoise, gzs = 0.0, [], 0.0, []
        for p in range(P):
            r, rg = orcg.svgp_elbo_value_and_grads(X, Y[:, p:p + 1], Zs[p], q_mu[:, p:p + 1], q_sqrt[p:p + 1], num_data=2000,
                                                   variance=vs[p], lengthscales=lss[p], noise_variance=0.2)
            rv += r; gq.append(rg["q_mu"]); gnoise += rg["noise_variance"]; gzs.append(rg["Z"])
        assert abs(v - rv) <= 1e-9 * abs(rv), (v, rv)
        assert abs(v - float(m.elbo((X, Y)).cpu())) <= 1e-9 * abs(v)
        np.testing.assert_allclose(g[m.q_mu], np.concatenate(gq, 1), rtol=0, atol=1e-8 * max(np.abs(q).max() for q in gq))
        u = m.likelihood.variance.unconstrained_variable
        np.testing.assert_allclose(np.asarray(g[m.likelihood.variance]).ravel(),
                                   (np.ravel(gnoise) * m.likelihood.variance.transform.forward_grad(u)).ravel(), rtol=1e-8)
        if separate_z:
            for p in range(P):
                zp = iv.inducing_variable_list[p].Z
                np.testing.assert_allclose(g[zp], gzs[p], rtol=0, atol=1e-8 * np.abs(gzs[p]).max())
        else:
            np.testing.assert_allclose(g[iv.inducing_variable.Z], sum(gzs), rtol=0, atol=1e-8 * np.abs(sum(gzs)).max())


@pytest.mark.parametrize("M,B,D,P", [(70, 300, 3, 2), (200, 640, 4, 1)])
def test_q_diag_svgp_elbo_and_grad_vs_autograd_oracle(gpu, M, B, D, P):
    """q_diag = True (q_sqrt [M, P] of standard deviations, svgp.py:90-148): value and every gradient of the whitened
    reverse pass against the autograd oracle; the oracle's q_diag forward is itself pinned to the NumPy oracle."""
    from gpflow_amd import gradients, ops
    from oracle import gp_oracle as orc
    X, Y, Z, q_mu, _, kw = _problem(M, B, D, P, 41)
    q = 0.3 + np.abs(np.random.default_rng(42).normal(size=(M, P)))
    t = ops.to_device
    F, g, info = gradients.svgp_elbo_and_grad(t(Z), t(X), t(Y), t(q_mu), t(q), jitter=1e-6, scale=1000.0 / B, mean_const=0.1, **kw)
    ops.check_info(info)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q, num_data=1000, mean=0.1, **kw)
    ref = orc.svgp_elbo(X, Y, Z, q_mu, q, num_data=1000, mean=0.1, whiten=True, **kw)
    assert abs(v - ref) <= 1e-10 * abs(ref)
    assert abs(float(F.cpu()[0]) - v) <= 1e-9 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, refg = g[name].cpu().numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(refg.shape), refg, rtol=0, atol=1e-8 * max(1.0, np.abs(refg).max()), err_msg=name)


def test_q_diag_model_gradients(gpu):
    """SVGP(q_diag=True).elbo_and_grad in the unconstrained space (softplus on the standard deviations) against finite
    differences of the model's own fused ELBO."""
    import gpflow_amd as gpflow
    X, Y, Z, q_mu, _, kw = _problem(40, 200, 2, 2, 43)
    q = 0.3 + np.abs(np.random.default_rng(44).normal(size=(40, 2)))
    m = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=kw["variance"], lengthscales=kw["lengthscales"]),
                           gpflow.likelihoods.Gaussian(kw["noise_variance"]), Z.copy(), q_mu=q_mu.copy(), q_sqrt=q.copy(), q_diag=True,
                           num_latent_gps=2, num_data=2000)
    v, g = m.elbo_and_grad((X, Y))
    assert abs(v - float(m.elbo((X, Y)).cpu())) <= 1e-9 * abs(v)
    h = 1e-5
    for par, idx in [(m.q_sqrt, (7, 1)), (m.q_sqrt, (31, 0)), (m.q_mu, (3, 1)), (m.kernel.lengthscales, (0,)), (m.inducing_variable.Z, (5, 1))]:
        u0 = np.array(par.unconstrained_variable, dtype=np.float64, copy=True)
        vals = []
        for d in (h, -h):
            u = u0.copy(); u[idx] += d
            par.assign_unconstrained(u)
            vals.append(float(m.elbo((X, Y)).cpu()))
        par.assign_unconstrained(u0)
        fd = (vals[0] - vals[1]) / (2 * h)
        got = float(np.asarray(g[par])[idx])
        assert abs(got - fd) <= 1e-5 * max(1.0, abs(fd)), (par.name, idx, got, fd)


# ----------------------------------------------------------------------------- Matern families (stationaries.py:254-313)
MATERN_VALUE_TOL = {"Matern12": 1e-7, "Matern32": 1e-9, "Matern52": 1e-9}
# (Matern12 VALUE: K_ii = variance * exp(-sqrt(max(r2_ii, 1e-36))) with r2_ii the rounding noise of the expansion formula, i.e.
#  1 - O(1e-8) or exactly 1 depending on the sign of that noise -- a different FMA order moves every diagonal entry by ~1e-8.)
MATERN_TOL = {"Matern12": 2e-6, "Matern32": 1e-8, "Matern52": 1e-8}
# (Matern12: d exp(-r)/dr2 = -exp(-r)/(2r) is unbounded at r -> 0; the autograd oracle differentiates the expansion formula,
#  whose diagonal r2_ii is rounding noise -- amplified by 1/r to ~1e-8 relative there.  The product writes the diagonal's
#  factor as exact zeros: r2_ii = 0 identically, no gradient flows through it.)


@pytest.mark.parametrize("family", ["SquaredExponential", "Matern12", "Matern32", "Matern52"])
def test_kernel_matrix_dr2_factor(gpu, family):
    """gpk_kernel_matrix_combine op 3: G .* (-2 dk/dr2) against the NumPy formulas; exact zeros on the diagonal of the
    symmetric form and where the 1e-36 clamp is active (duplicate rows)."""
    import fake_ops
    from gpflow_amd import ops
    rng = np.random.default_rng(5)
    for n1, n2, d in [(70, 130, 3), (256, 512, 8), (1, 5, 1)]:
        A, Bm, G = rng.normal(size=(n1, d)), rng.normal(size=(n2, d)), rng.normal(size=(n1, n2))
        if n2 > 3:
            Bm[2] = A[0]                      # an exactly coincident pair: r2 = rounding noise, possibly clamped
        ls = 0.7 + 0.1 * np.arange(d)
        got = ops.kernel_matrix_combine(ops.to_device(A), ops.to_device(Bm), ops.to_device(G), op="dr2", variance=1.7,
                                        lengthscales=ls, family=family).cpu().numpy()
        ref = fake_ops._dr2(A, Bm, 1.7, ls, family) * G
        mask = np.ones_like(ref, dtype=bool)
        if n2 > 3:
            mask[0, 2] = False                # (noise-level r2: either side of the clamp; not comparable)
        np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-11, atol=1e-13)
        S = rng.normal(size=(n1, n1))
        got = ops.kernel_matrix_combine(ops.to_device(A), None, ops.to_device(S), op="dr2", variance=1.7, lengthscales=ls,
                                        family=family).cpu().numpy()
        ref = fake_ops._dr2(A, A, 1.7, ls, family) * S
        np.fill_diagonal(ref, 0.0)
        assert np.all(np.diag(got) == 0.0)
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("family", ["Matern12", "Matern32", "Matern52"])
@pytest.mark.parametrize("whiten", [True, False])
def test_matern_svgp_elbo_and_grad_vs_autograd_oracle(gpu, family, whiten):
    from gpflow_amd import gradients, ops
    M, B, D, P = 200, 520, 4, 2
    X, Y, Z, q_mu, q_sqrt, kw = _problem(M, B, D, P, 52, True)
    t = ops.to_device
    fn = gradients.svgp_elbo_and_grad if whiten else gradients.svgp_elbo_and_grad_unwhitened
    F, g, info = fn(t(Z), t(X), t(Y), t(q_mu), t(q_sqrt), jitter=1e-6, scale=1000.0 / B, mean_const=0.1, family=family, **kw)
    ops.check_info(info)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, num_data=1000, mean=0.1, whiten=whiten, family=family, **kw)
    assert abs(float(F.cpu()[0]) - v) <= MATERN_VALUE_TOL[family] * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, ref = g[name].cpu().numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=MATERN_TOL[family] * max(1.0, np.abs(ref).max()),
                                   err_msg=f"{family} {name}")


@pytest.mark.parametrize("family", ["Matern12", "Matern32", "Matern52"])
def test_matern_gpr_and_sgpr_gradients_vs_autograd_oracle(gpu, family):
    from gpflow_amd import gradients, ops
    rng = np.random.default_rng(53)
    N, M, D, P = 700, 130, 3, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P)); Z = rng.normal(size=(M, D))
    kw = dict(variance=1.4, lengthscales=np.sqrt(D) * (0.8 + 0.05 * np.arange(D)), noise_variance=0.15)
    t = ops.to_device
    F, g, info = gradients.gpr_lml_and_grad(t(X), t(Y), mean_const=0.2, family=family, **kw)
    ops.check_info(info)
    v, go = orcg.gpr_lml_value_and_grads(X, Y, mean=0.2, family=family, **kw)
    assert abs(float(F.cpu()[0]) - v) <= MATERN_VALUE_TOL[family] * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "mean_const"):
        got, ref = g[name].cpu().numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=MATERN_TOL[family] * max(1.0, np.abs(ref).max()),
                                   err_msg=f"gpr {family} {name}")
    F, g, info = gradients.sgpr_elbo_and_grad(t(Z), t(X), t(Y), jitter=1e-6, mean_const=0.2, family=family, **kw)
    ops.check_info(info)
    v, go = orcg.sgpr_elbo_value_and_grads(X, Y, Z, mean=0.2, family=family, **kw)
    assert abs(float(F.cpu()[0]) - v) <= MATERN_VALUE_TOL[family] * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "mean_const"):
        got, ref = g[name].cpu().numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=MATERN_TOL[family] * max(1.0, np.abs(ref).max()),
                                   err_msg=f"sgpr {family} {name}")


def test_matern_models_train_through_the_public_surface(gpu):
    """Matern kernels through the model entry points: GPR(Matern32, active_dims).log_marginal_likelihood_and_grad and
    SGPR(Matern52, active_dims).objective_and_grad against the autograd oracle on the sliced inputs (chain rule to the
    unconstrained variables included), the Scipy optimizer on a Matern52 GPR, the SVGP trainer on a Matern32 SVGP."""
    import gpflow_amd as gpflow
    rng = np.random.default_rng(54)
    N, D = 300, 4
    X = rng.normal(size=(N, D)); Y = np.sin(X[:, [1]] + X[:, [3]]) + 0.1 * rng.normal(size=(N, 1))
    dims, ls = [3, 1], np.array([0.9, 1.4])
    k = gpflow.kernels.Matern32(variance=1.3, lengthscales=ls, active_dims=dims)
    m = gpflow.models.GPR((X, Y), k, noise_variance=0.2)
    v, g = m.log_marginal_likelihood_and_grad()
    rv, rg = orcg.gpr_lml_value_and_grads(X[:, dims], Y, variance=1.3, lengthscales=ls, noise_variance=0.2, family="Matern32")
    assert abs(v - rv) <= 1e-9 * abs(rv) and abs(v - float(m.log_marginal_likelihood().cpu())) <= 1e-9 * abs(v)
    u = k.lengthscales.unconstrained_variable
    np.testing.assert_allclose(np.asarray(g[k.lengthscales]).ravel(), (rg["lengthscales"] * k.lengthscales.transform.forward_grad(u)).ravel(),
                               rtol=1e-7)
    # SGPR, Matern52, active_dims: Z gradient lands in the active columns only
    Z = rng.normal(size=(40, D))
    k2 = gpflow.kernels.Matern52(variance=1.1, lengthscales=ls, active_dims=dims)
    s = gpflow.models.SGPR((X, Y), k2, Z.copy(), noise_variance=0.2)
    v, g = s.objective_and_grad()
    rv, rg = orcg.sgpr_elbo_value_and_grads(X[:, dims], Y, Z[:, dims], variance=1.1, lengthscales=ls, noise_variance=0.2, family="Matern52")
    assert abs(v - rv) <= 1e-9 * abs(rv)
    gz = np.asarray(g[s.inducing_variable.Z])
    np.testing.assert_allclose(gz[:, dims], rg["Z"], rtol=0, atol=1e-8 * np.abs(rg["Z"]).max())
    assert np.all(gz[:, [0, 2]] == 0.0)
    # a NESTED combination (refused until the end of round 5): the value of the reverse pass is the forward LML
    mn = gpflow.models.GPR((X, Y), (k + gpflow.kernels.SquaredExponential()) * gpflow.kernels.Matern12())
    vn, gn = mn.log_marginal_likelihood_and_grad()
    assert abs(vn - float(mn.log_marginal_likelihood().cpu())) <= 1e-9 * abs(vn) and len(gn) == 7
    # Scipy on a Matern52 GPR improves the LML
    m2 = gpflow.models.GPR((X, Y), gpflow.kernels.Matern52(lengthscales=np.ones(D)), noise_variance=1.0)
    before = float(m2.log_marginal_likelihood().cpu())
    gpflow.optimizers.Scipy().minimize(m2, options=dict(maxiter=15))
    assert float(m2.log_marginal_likelihood().cpu()) > before + 10.0
    # the device-resident trainer on a Matern32 SVGP
    sv = gpflow.models.SVGP(gpflow.kernels.Matern32(lengthscales=np.ones(D)), gpflow.likelihoods.Gaussian(0.5), Z.copy(), num_data=N)
    tr = gpflow.training.SVGPTrainer(sv, learning_rate=0.05)
    first = float(tr.step((X, Y)).cpu()[0])
    for _ in range(30):
        last = tr.step((X, Y))
    assert float(last.cpu()[0]) > first
    tr.sync_to_model()
    assert abs(float(sv.elbo((X, Y)).cpu()) - float(tr.step((X, Y)).cpu()[0])) <= 1e-8 * abs(first)


@pytest.mark.parametrize("M,B,D,P,family", [(70, 300, 3, 2, "SquaredExponential"), (200, 640, 4, 1, "Matern52")])
def test_unwhitened_q_diag_svgp_elbo_and_grad_vs_autograd_oracle(gpu, M, B, D, P, family):
    """whiten=False with q_diag=True: value and every gradient against the autograd oracle (whose forward is pinned to the
    NumPy oracle for this combination in tests/test_gradients_cpu.py), then through SVGP.elbo_and_grad against the model's
    own fused forward."""
    import gpflow_amd as gpflow
    from gpflow_amd import gradients, ops
    X, Y, Z, q_mu, _, kw = _problem(M, B, D, P, 61)
    q = 0.3 + np.abs(np.random.default_rng(62).normal(size=(M, P)))
    t = ops.to_device
    F, g, info = gradients.svgp_elbo_and_grad_unwhitened(t(Z), t(X), t(Y), t(q_mu), t(q), jitter=1e-6, scale=1000.0 / B,
                                                         mean_const=0.1, family=family, **kw)
    ops.check_info(info)
    v, go = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q, num_data=1000, mean=0.1, whiten=False, family=family, **kw)
    assert abs(float(F.cpu()[0]) - v) <= 1e-9 * abs(v)
    for name in ("variance", "lengthscales", "noise_variance", "Z", "q_mu", "q_sqrt", "mean_const"):
        got, refg = g[name].cpu().numpy(), np.asarray(go[name])
        np.testing.assert_allclose(got.reshape(refg.shape), refg, rtol=0, atol=1e-8 * max(1.0, np.abs(refg).max()), err_msg=name)
    kern = getattr(gpflow.kernels, family)(variance=kw["variance"], lengthscales=kw["lengthscales"])
    m = gpflow.models.SVGP(kern, gpflow.likelihoods.Gaussian(kw["noise_variance"]), Z.copy(), q_mu=q_mu.copy(), q_sqrt=q.copy(),
                           q_diag=True, whiten=False, num_latent_gps=P, num_data=1000 * 1, mean_function=gpflow.mean_functions.Constant(0.1))
    mv, mg = m.elbo_and_grad((X, Y))
    scale_fix = (1000.0 / B)   # (the model scales by num_data / B itself)
    assert abs(mv - float(m.elbo((X, Y)).cpu())) <= 1e-9 * abs(mv) and abs(mv - v) <= 1e-9 * abs(v), (mv, v, scale_fix)
    np.testing.assert_allclose(mg[m.q_mu], go["q_mu"], rtol=0, atol=1e-8 * np.abs(go["q_mu"]).max())


@pytest.mark.parametrize("whiten", [True, False])
def test_trainer_with_q_diag_and_active_dims(gpu, whiten):
    """The device-resident trainer on a model with a diagonal q_sqrt (softplus-constrained standard deviations, chained on
    the device) and a kernel with `active_dims`: the value it returns is the model's own ELBO, its FIRST Adam step moves
    every variable by lr * sign(dELBO/du) (bias-corrected Adam from zero moments) with the gradient of `SVGP.elbo_and_grad`,
    inducing-point columns outside `active_dims` do not move, and training raises the ELBO."""
    import gpflow_amd as gpflow
    rng = np.random.default_rng(71)
    N, D, M, P = 260, 4, 30, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X[:, [2]] + X[:, [0]]) + 0.1 * rng.normal(size=(N, P))
    Z = rng.normal(size=(M, D)); q_mu = 0.2 * rng.normal(size=(M, P)); q = 0.4 + np.abs(rng.normal(size=(M, P)))
    dims = [2, 0]
    def make():
        k = gpflow.kernels.Matern32(variance=1.2, lengthscales=np.array([0.9, 1.3]), active_dims=dims)
        return gpflow.models.SVGP(k, gpflow.likelihoods.Gaussian(0.3), Z.copy(), q_mu=q_mu.copy(), q_sqrt=q.copy(), q_diag=True,
                                  whiten=whiten, num_latent_gps=P, num_data=5 * N)
    m = make()
    v, g = m.elbo_and_grad((X, Y))
    lr = 0.01
    tr = gpflow.training.SVGPTrainer(m, learning_rate=lr)
    u0 = np.array(m.q_sqrt.unconstrained_variable, dtype=np.float64, copy=True)
    F0 = float(tr.step((X, Y)).cpu()[0])
    assert abs(F0 - v) <= 1e-9 * abs(v)
    tr.sync_to_model()
    du = np.asarray(m.q_sqrt.unconstrained_variable) - u0
    gq = np.asarray(g[m.q_sqrt]).reshape(u0.shape)
    big = np.abs(gq) > 1e-3 * np.abs(gq).max()
    np.testing.assert_allclose(du[big], lr * np.sign(gq[big]), rtol=1e-4)
    dz = m.inducing_variable.Z.numpy() - Z
    assert np.all(dz[:, [1, 3]] == 0.0) and np.abs(dz[:, dims]).max() > 0.5 * lr
    for _ in range(40):
        last = tr.step((X, Y))
    assert float(last.cpu()[0]) > F0
    tr.sync_to_model()
    assert abs(float(m.elbo((X, Y)).cpu()) - float(tr.step((X, Y)).cpu()[0])) <= 1e-8 * abs(F0)


def test_parameter_priors_enter_value_and_gradient(gpu):
    """MAP estimation (gpflow/models/model.py:47-76, gpflow/base.py:201-224): with priors on trainable parameters the
    objective handed to the optimisers is the log POSTERIOR density -- objective + sum of log prior densities -- and its
    gradient carries the priors' part, for a prior on the constrained value (Gamma on lengthscales / variance) and for one
    on the unconstrained value (Normal, with the log|Jacobian| of the transform).  GPR, SGPR and SVGP; checked against
    central differences of the models' own `log_posterior_density` in the unconstrained space; a user prior object without
    `grad_log_prob` goes through the numerical fallback; Scipy's MAP fit differs from the ML fit."""
    import gpflow_amd as gpflow
    from gpflow_amd import priors
    from gpflow_amd.base import PriorOn
    rng = np.random.default_rng(23)
    X = rng.uniform(-2, 2, size=(60, 2)); Y = np.sin(2 * X[:, :1]) + 0.1 * rng.normal(size=(60, 1))

    class OnlyLogProb:   # e.g. a tfp distribution wrapped by the user: no derivative supplied
        def log_prob(self, x):
            return -0.5 * ((np.asarray(x) - 0.7) / 0.3) ** 2

    def put_priors(m):
        m.kernel.lengthscales.prior = priors.Gamma(2.0, 3.0)
        m.kernel.variance.prior = priors.LogNormal(0.1, 0.8)
        m.likelihood.variance.prior = priors.Normal(-1.0, 2.0)
        m.likelihood.variance.prior_on = PriorOn.UNCONSTRAINED
        return m

    def check(m, objective, value_fn):
        v, g = objective()
        lp = m.log_prior_density()
        assert abs(lp) > 1e-3
        assert abs(v - (value_fn() + lp)) <= 1e-9 * max(1.0, abs(v))
        h = 1e-5
        for par in (m.kernel.lengthscales, m.kernel.variance, m.likelihood.variance):
            u0 = np.array(par.unconstrained_variable, dtype=np.float64, copy=True)
            for idx in np.ndindex(*u0.shape) if u0.shape else [()]:
                vals = []
                for d in (h, -h):
                    u = u0.copy(); u[idx] += d
                    par.assign_unconstrained(u)
                    vals.append(value_fn() + m.log_prior_density())
                par.assign_unconstrained(u0)
                fd = (vals[0] - vals[1]) / (2 * h)
                got = float(np.asarray(g[par])[idx])
                assert abs(got - fd) <= 2e-5 * max(1.0, abs(fd)), (par.name, idx, got, fd)

    gpr = put_priors(gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(lengthscales=[0.9, 1.3]), noise_variance=0.3))
    check(gpr, gpr.objective_and_grad, lambda: float(gpr.log_marginal_likelihood().cpu()))
    sg = put_priors(gpflow.models.SGPR((X, Y), gpflow.kernels.SquaredExponential(lengthscales=[0.9, 1.3]), X[:9].copy(), noise_variance=0.3))
    check(sg, sg.objective_and_grad, lambda: float(sg.elbo().cpu()))
    sv, Xs, Ys = _small_model(20, 60, 2, 1, 5)
    put_priors(sv)
    check(sv, lambda: sv.elbo_and_grad((Xs, Ys)), lambda: float(sv.elbo((Xs, Ys)).cpu()))
    # numerical fallback for a prior without grad_log_prob
    gpr.kernel.variance.prior = OnlyLogProb()
    check(gpr, gpr.objective_and_grad, lambda: float(gpr.log_marginal_likelihood().cpu()))
    # the MAP optimum is not the ML optimum, and training_loss == -log posterior there
    ml = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(lengthscales=[0.9, 1.3]), noise_variance=0.3)
    gpflow.optimizers.Scipy().minimize(ml, options=dict(maxiter=200))
    res = gpflow.optimizers.Scipy().minimize(gpr, options=dict(maxiter=200))
    assert abs(res.fun - float(gpr.training_loss())) <= 1e-8 * abs(res.fun)
    assert np.max(np.abs(gpr.kernel.lengthscales.numpy() - ml.kernel.lengthscales.numpy())) > 1e-3


@pytest.mark.parametrize("q_diag", [False, True])
def test_trainer_priors_on_device_resident_variables(gpu, q_diag):
    """MAP training with priors on the variables SVGPTrainer keeps in HBM (models/model.py:47-76 sums the priors of EVERY trainable
    parameter, Z / q_mu / q_sqrt included; base.py:201-224).  The first bias-corrected Adam step moves every entry by
    lr * sign(d log-posterior / du): checked against ELBO gradient + prior gradient formed independently on the host; the
    reported objective is the log posterior density; a user prior object without torch methods takes the host route."""
    import gpflow_amd as gpflow
    from gpflow_amd import priors
    from gpflow_amd.base import PriorOn
    rng = np.random.default_rng(83)
    N, D, M, P = 200, 2, 24, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X[:, :1]) + 0.1 * rng.normal(size=(N, P))
    Z = rng.normal(size=(M, D)); q_mu = 0.2 * rng.normal(size=(M, P))
    qs = 0.4 + np.abs(rng.normal(size=(M, P))) if q_diag else np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)

    class OnlyLogProb:   # a user prior: NumPy log_prob only (tfp-style object)
        def log_prob(self, x):
            return -0.5 * (np.asarray(x) / 0.7) ** 2

    def make():
        m = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.2, lengthscales=[0.9, 1.3]), gpflow.likelihoods.Gaussian(0.3),
                               Z.copy(), q_mu=q_mu.copy(), q_sqrt=qs.copy(), q_diag=q_diag, num_latent_gps=P, num_data=5 * N)
        m.inducing_variable.Z.prior = priors.Normal(0.3, 0.05)          # strong enough to flip signs of the ELBO gradient
        m.q_mu.prior = OnlyLogProb()
        m.q_sqrt.prior = priors.Normal(-1.0, 0.05) if q_diag else priors.Normal(0.0, 0.02)   # (q_diag: on the unconstrained value)
        if q_diag:
            m.q_sqrt.prior_on = PriorOn.UNCONSTRAINED
        m.kernel.variance.prior = priors.LogNormal(0.1, 0.8)
        return m
    m = make()
    plain = make()
    for par in plain.trainable_parameters:
        par.prior = None
    v, g = plain.elbo_and_grad((X, Y))                                     # ELBO and its gradient, no priors
    # the priors' part, on the host, in the space of the trainer's device variables
    Zg = priors.grad_log_prob(m.inducing_variable.Z.prior, Z)
    mug = priors.grad_log_prob(m.q_mu.prior, q_mu)
    if q_diag:
        u = np.asarray(m.q_sqrt.unconstrained_variable, dtype=np.float64)
        sig = 1.0 / (1.0 + np.exp(-u))
        qg = priors.grad_log_prob(m.q_sqrt.prior, u) - (1.0 - sig)
        lp_q = float(np.sum(m.q_sqrt.prior.log_prob(u)) - np.sum(np.log(sig)))
        g_q = np.asarray(g[plain.q_sqrt]).reshape(u.shape) + qg
    else:
        low = np.tril(qs)
        qg = np.tril(priors.grad_log_prob(m.q_sqrt.prior, low))
        lp_q = float(np.sum(m.q_sqrt.prior.log_prob(low)))
        from gpflow_amd.base import FillTriangular
        g_q = FillTriangular().forward(np.asarray(g[plain.q_sqrt]).reshape(P, -1)) + qg     # back to [P, M, M]
    lp = float(np.sum(m.inducing_variable.Z.prior.log_prob(Z)) + np.sum(m.q_mu.prior.log_prob(q_mu))) + lp_q \
        + m.kernel.variance.log_prior_density()
    lr = 0.01
    tr = gpflow.training.SVGPTrainer(m, learning_rate=lr)
    F0 = float(tr.step((X, Y)).cpu()[0])
    assert abs(F0 - (v + lp)) <= 1e-9 * abs(v + lp), (F0, v, lp)
    tr.sync_to_model()

    def moved(new, old, grad):
        big = np.abs(grad) > 1e-3 * np.abs(grad).max()
        np.testing.assert_allclose((new - old)[big], lr * np.sign(grad[big]), rtol=1e-4)
    gZ = np.asarray(g[plain.inducing_variable.Z]) + Zg
    assert (np.sign(gZ) != np.sign(np.asarray(g[plain.inducing_variable.Z]))).any()      # the prior really changes the step
    moved(m.inducing_variable.Z.numpy(), Z, gZ)
    moved(m.q_mu.numpy(), q_mu, np.asarray(g[plain.q_mu]) + mug)
    if q_diag:
        moved(np.asarray(m.q_sqrt.unconstrained_variable), u, g_q)
    else:
        tri = np.tril(np.ones((M, M), dtype=bool))
        moved(m.q_sqrt.numpy()[:, tri], qs[:, tri], g_q[:, tri])
    for _ in range(30):
        last = tr.step((X, Y))
    assert float(last.cpu()[0]) > F0


@pytest.mark.parametrize("op", ["add", "mul"])
def test_sum_and_product_kernels_in_the_reverse_pass(gpu, op):
    """Sum / Product of stationary kernels (gpflow/kernels/base.py:216-220, 305-315; the reference differentiates the
    tf.add_n / tf.multiply of the member matrices): GPR.log_marginal_likelihood and the whitened SVGP.elbo with their
    gradients w.r.t. EVERY member's variance and lengthscales (plus noise, Z, q_mu, q_sqrt) against torch autograd over the
    restated kernels; through the model surface in the unconstrained space, and Scipy improves the objective."""
    import gpflow_amd as gpflow
    from gpflow_amd import gradients, ops
    rng = np.random.default_rng(31)
    N, D, M, P = 260, 3, 70, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P))
    Z = X[:M] + 0.05 * rng.normal(size=(M, D))
    q_mu = 0.2 * rng.normal(size=(M, P)); q_sqrt = np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)
    members = [("SquaredExponential", 1.2, np.array([0.9, 1.1, 1.3])), ("Matern32", 0.7, np.array(0.8)), ("Matern52", 0.9, np.array([1.5, 0.7, 1.0]))]
    spec = gradients.KernelSpec(members, op)
    t = ops.to_device

    def chk(got, ref, tol=1e-8):
        got = np.asarray(got.cpu().numpy() if hasattr(got, "cpu") else got, dtype=np.float64).reshape(np.shape(ref))
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (np.abs(got - ref).max(), np.abs(ref).max())
    # GPR
    F, g, info = gradients.gpr_lml_and_grad(t(X), t(Y[:, :1]), noise_variance=0.2, mean_const=0.1, kernel_spec=spec)
    rv, rg = orcg.combination_value_and_grads("gpr", X, Y[:, :1], members, op, noise_variance=0.2, mean=0.1)
    assert int(info.cpu()[0]) == 0 and abs(float(F.cpu()[0]) - rv) <= 1e-9 * abs(rv)
    chk(g["variance"], rg["variance"]); chk(g["noise_variance"], rg["noise_variance"])
    for i in range(3):
        chk(g["lengthscales"][i], rg["lengthscales"][i])
    # SVGP (whitened)
    F, g, info = gradients.svgp_elbo_and_grad(t(Z), t(X), t(Y), t(q_mu), t(q_sqrt), noise_variance=0.2, jitter=1e-6, scale=5.0,
                                              kernel_spec=spec)
    rv, rg = orcg.combination_value_and_grads("svgp", X, Y, members, op, noise_variance=0.2, Z=Z, q_mu=q_mu, q_sqrt=q_sqrt, num_data=5 * N)
    assert int(info.cpu()[0]) == 0 and abs(float(F.cpu()[0]) - rv) <= 1e-9 * abs(rv)
    chk(g["variance"], rg["variance"]); chk(g["noise_variance"], rg["noise_variance"]); chk(g["Z"], rg["Z"]); chk(g["q_mu"], rg["q_mu"])
    chk(np.tril(g["q_sqrt"].cpu().numpy()), np.tril(rg["q_sqrt"]))
    for i in range(3):
        chk(g["lengthscales"][i], rg["lengthscales"][i])
    # model surface: unconstrained gradients of every member parameter; then Scipy
    ks = [gpflow.kernels.SquaredExponential(variance=1.2, lengthscales=[0.9, 1.1, 1.3]), gpflow.kernels.Matern32(variance=0.7, lengthscales=0.8)]
    kc = ks[0] + ks[1] if op == "add" else ks[0] * ks[1]
    m = gpflow.models.GPR((X, Y[:, :1]), kc, noise_variance=0.2)
    v, gm = m.objective_and_grad()
    assert abs(v - float(m.log_marginal_likelihood().cpu())) <= 1e-9 * abs(v)
    assert set(gm) == {ks[0].variance, ks[0].lengthscales, ks[1].variance, ks[1].lengthscales, m.likelihood.variance}
    h = 1e-5
    for par, idx in [(ks[0].variance, ()), (ks[0].lengthscales, (1,)), (ks[1].variance, ()), (ks[1].lengthscales, ())]:
        u0 = np.array(par.unconstrained_variable, dtype=np.float64, copy=True)
        vals = []
        for dlt in (h, -h):
            u = u0.copy(); u[idx] += dlt
            par.assign_unconstrained(u)
            vals.append(float(m.log_marginal_likelihood().cpu()))
        par.assign_unconstrained(u0)
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(float(np.asarray(gm[par])[idx]) - fd) <= 2e-5 * max(1.0, abs(fd)), (par.name, fd)
    res = gpflow.optimizers.Scipy().minimize(m, options=dict(maxiter=20))
    assert -res.fun > v
    sv = gpflow.models.SVGP(kc, gpflow.likelihoods.Gaussian(0.2), Z.copy(), q_mu=q_mu, q_sqrt=q_sqrt, num_latent_gps=P, num_data=5 * N)
    v2, g2 = sv.elbo_and_grad((X, Y))
    assert abs(v2 - float(sv.elbo((X, Y)).cpu())) <= 1e-9 * abs(v2) and ks[1].lengthscales in g2 and sv.inducing_variable.Z in g2
    # a DIAGONAL q_sqrt under the combination (svgp.py:90-148 with q_diag = True; refused until round 5), whitened and not:
    # the reverse pass against autograd over the restated model, then the model surface against central differences of its own ELBO
    qd = 0.4 + 0.2 * np.abs(rng.normal(size=(M, P)))
    for wh, name, fn in ((True, "svgp", gradients.svgp_elbo_and_grad), (False, "svgp_unwhitened", gradients.svgp_elbo_and_grad_unwhitened)):
        F, g, info = fn(t(Z), t(X), t(Y), t(q_mu), t(qd), noise_variance=0.2, jitter=1e-6, scale=5.0, kernel_spec=spec)
        rv, rg = orcg.combination_value_and_grads(name, X, Y, members, op, noise_variance=0.2, Z=Z, q_mu=q_mu, q_sqrt=qd, num_data=5 * N)
        assert int(info.cpu()[0]) == 0 and abs(float(F.cpu()[0]) - rv) <= 1e-9 * abs(rv)
        for k in ("variance", "noise_variance", "Z", "q_mu", "q_sqrt"):
            chk(g[k], rg[k], 1e-7 if not wh else 1e-8)
        for i in range(3):
            chk(g["lengthscales"][i], rg["lengthscales"][i], 1e-7 if not wh else 1e-8)
        sd = gpflow.models.SVGP(kc, gpflow.likelihoods.Gaussian(0.2), Z.copy(), q_mu=q_mu, q_sqrt=qd, q_diag=True, whiten=wh,
                                num_latent_gps=P, num_data=5 * N)
        v3, g3 = sd.elbo_and_grad((X, Y))
        assert abs(v3 - float(sd.elbo((X, Y)).cpu())) <= 1e-9 * abs(v3) and sd.q_sqrt in g3 and ks[0].lengthscales in g3
        par = sd.q_sqrt
        u0 = np.array(par.unconstrained_variable, dtype=np.float64, copy=True)
        vals = []
        for dlt in (1e-5, -1e-5):
            u = u0.copy(); u[3, 1] += dlt
            par.assign_unconstrained(u)
            vals.append(float(sd.elbo((X, Y)).cpu()))
        par.assign_unconstrained(u0)
        fd = (vals[0] - vals[1]) / 2e-5
        assert abs(float(np.asarray(g3[par])[3, 1]) - fd) <= 2e-5 * max(1.0, abs(fd)), fd


@pytest.mark.parametrize("op", ["add", "mul"])
def test_combinations_over_different_active_dims_and_under_sgpr_and_the_unwhitened_svgp(gpu, op):
    """What round 4 still refused (kernels/base.py:90-109, 216-220, 283-329; models/sgpr.py:181-290): Sum / Product kernels whose
    members see DIFFERENT input columns (each member slices for itself; its input gradient is scattered back), and kernel
    combinations under SGPR and under the un-whitened SVGP -- values and every gradient against torch autograd over the restated
    kernels, then through the model surface."""
    import gpflow_amd as gpflow
    from gpflow_amd import gradients, ops
    rng = np.random.default_rng(37)
    N, D, M, P = 230, 4, 60, 2
    X = rng.normal(size=(N, D)); Y = np.sin(X[:, :2].sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P))
    Z = X[:M] + 0.05 * rng.normal(size=(M, D))
    q_mu = 0.2 * rng.normal(size=(M, P)); q_sqrt = np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)
    members = [("SquaredExponential", 1.2, np.array([0.9, 1.1])), ("Matern32", 0.7, np.array(0.8)), ("Matern52", 0.9, np.array([1.5, 0.7, 1.0, 1.2]))]
    cols = [[0, 2], [1, 2, 3], None]
    spec = gradients.KernelSpec(members, op, cols=cols)
    t = ops.to_device

    def chk(got, ref, tol=1e-8):
        got = np.asarray(got.cpu().numpy() if hasattr(got, "cpu") else got, dtype=np.float64).reshape(np.shape(ref))
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (np.abs(got - ref).max(), np.abs(ref).max())

    def chk_kernel(g, rg):
        chk(g["variance"], rg["variance"]); chk(g["noise_variance"], rg["noise_variance"])
        for i in range(3):
            chk(g["lengthscales"][i], rg["lengthscales"][i])
    # GPR and the whitened SVGP with per-member columns
    F, g, info = gradients.gpr_lml_and_grad(t(X), t(Y[:, :1]), noise_variance=0.2, mean_const=0.1, kernel_spec=spec)
    rv, rg = orcg.combination_value_and_grads("gpr", X, Y[:, :1], members, op, noise_variance=0.2, mean=0.1, cols=cols)
    assert int(info.cpu()[0]) == 0 and abs(float(F.cpu()[0]) - rv) <= 1e-9 * abs(rv)
    chk_kernel(g, rg)
    common = dict(noise_variance=0.2, Z=Z, q_mu=q_mu, q_sqrt=q_sqrt, num_data=5 * N, cols=cols)
    for name, fn in (("svgp", gradients.svgp_elbo_and_grad), ("svgp_unwhitened", gradients.svgp_elbo_and_grad_unwhitened)):
        F, g, info = fn(t(Z), t(X), t(Y), t(q_mu), t(q_sqrt), noise_variance=0.2, jitter=1e-6, scale=5.0, kernel_spec=spec)
        rv, rg = orcg.combination_value_and_grads(name, X, Y, members, op, **common)
        tol = 1e-8 if name == "svgp" else 1e-7     # (one more kappa(Lm) in the un-whitened chain)
        assert int(info.cpu()[0]) == 0 and abs(float(F.cpu()[0]) - rv) <= 1e-9 * abs(rv)
        chk(g["variance"], rg["variance"], tol); chk(g["noise_variance"], rg["noise_variance"], tol)
        chk(g["Z"], rg["Z"], tol); chk(g["q_mu"], rg["q_mu"], tol)
        chk(np.tril(g["q_sqrt"].cpu().numpy()), np.tril(rg["q_sqrt"]), tol)
        for i in range(3):
            chk(g["lengthscales"][i], rg["lengthscales"][i], tol)
    # SGPR over the combination
    F, g, info = gradients.sgpr_elbo_and_grad(t(Z), t(X), t(Y), noise_variance=0.2, jitter=1e-6, mean_const=0.1, kernel_spec=spec)
    rv, rg = orcg.combination_value_and_grads("sgpr", X, Y, members, op, noise_variance=0.2, Z=Z, mean=0.1, cols=cols)
    assert int(info.cpu()[0]) == 0 and abs(float(F.cpu()[0]) - rv) <= 1e-9 * abs(rv)
    chk_kernel(g, rg); chk(g["Z"], rg["Z"])
    # model surface: members with active_dims (an index list and a slice), a Parameter shared by two members
    k0 = gpflow.kernels.SquaredExponential(variance=1.2, lengthscales=[0.9, 1.1], active_dims=[0, 2])
    k1 = gpflow.kernels.Matern32(variance=0.7, lengthscales=0.8, active_dims=slice(1, 4))
    kc = k0 + k1 if op == "add" else k0 * k1
    mem2, cols2 = members[:2], cols[:2]
    m = gpflow.models.GPR((X, Y[:, :1]), kc, noise_variance=0.2)
    v, gm = m.objective_and_grad()
    rv, rg = orcg.combination_value_and_grads("gpr", X, Y[:, :1], mem2, op, noise_variance=0.2, cols=cols2)
    assert abs(v - rv) <= 1e-9 * abs(rv) and abs(v - float(m.log_marginal_likelihood().cpu())) <= 1e-9 * abs(v)
    chk(gm[k0.lengthscales] / k0.lengthscales.transform.forward_grad(k0.lengthscales.unconstrained_variable), rg["lengthscales"][0])
    sg = gpflow.models.SGPR((X, Y), kc, Z.copy(), noise_variance=0.2)
    v, gs = sg.objective_and_grad()
    rv, rg = orcg.combination_value_and_grads("sgpr", X, Y, mem2, op, noise_variance=0.2, Z=Z, cols=cols2)
    assert abs(v - rv) <= 1e-9 * abs(rv) and abs(v - float(sg.elbo().cpu())) <= 1e-9 * abs(v)
    chk(gs[sg.inducing_variable.Z], rg["Z"])
    mu, var = sg.predict_f(X[:7])
    assert mu.shape == (7, P) and bool((var > 0).all())
    for wh in (True, False):
        sv = gpflow.models.SVGP(kc, gpflow.likelihoods.Gaussian(0.2), Z.copy(), q_mu=q_mu, q_sqrt=q_sqrt, num_latent_gps=P,
                                num_data=5 * N, whiten=wh)
        v2, g2 = sv.elbo_and_grad((X, Y))
        rv, rg = orcg.combination_value_and_grads("svgp" if wh else "svgp_unwhitened", X, Y, mem2, op, noise_variance=0.2, Z=Z,
                                                  q_mu=q_mu, q_sqrt=q_sqrt, num_data=5 * N, cols=cols2)
        assert abs(v2 - rv) <= 1e-9 * abs(rv) and abs(v2 - float(sv.elbo((X, Y)).cpu())) <= 1e-9 * abs(v2)
        chk(g2[sv.inducing_variable.Z], rg["Z"], 1e-7)
    # a Parameter that appears in two members (k + k): its gradient is the SUM over the members (what autodiff returns)
    ksh = gpflow.kernels.SquaredExponential(variance=0.6, lengthscales=1.1)
    kk = ksh + ksh if op == "add" else ksh * ksh
    m2 = gpflow.models.GPR((X, Y[:, :1]), kk, noise_variance=0.2)
    _, gdup = m2.objective_and_grad()
    _, rg = orcg.combination_value_and_grads("gpr", X, Y[:, :1], [("SquaredExponential", 0.6, np.array(1.1))] * 2, op, noise_variance=0.2)
    chk(gdup[ksh.variance] / ksh.variance.transform.forward_grad(ksh.variance.unconstrained_variable), rg["variance"].sum())


@pytest.mark.parametrize("shared", [False, True])
def test_trainer_with_kernel_combinations(gpu, shared):
    """SVGPTrainer under a kernel combination (refused until the end of round 5): (SquaredExponential + Matern32[dim 1]) * Matern52, and
    k + k with SHARED Parameters (one host entry per distinct Parameter; it collects the sum of its members' gradients).  The first objective
    is SVGP.elbo_and_grad's, the first Adam step moves every kernel parameter by the learning rate along that gradient, the bound improves."""
    import gpflow_amd as gpflow
    from gpflow_amd import training
    rng = np.random.default_rng(3)
    N, M, P = 200, 20, 2
    X = rng.normal(size=(N, 3)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P))
    Z = X[:M].copy(); q_mu = 0.2 * rng.normal(size=(M, P)); qs = np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)
    for wh in (True, False):
        k0 = gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.9, 1.1, 1.3])
        kern = (k0 + k0) if shared else (k0 + gpflow.kernels.Matern32(variance=0.6, lengthscales=0.8, active_dims=[1])) \
            * gpflow.kernels.Matern52(variance=0.9, lengthscales=1.2)
        m = gpflow.models.SVGP(kern, gpflow.likelihoods.Gaussian(0.2), Z.copy(), q_mu=q_mu, q_sqrt=qs, whiten=wh, num_latent_gps=P, num_data=5 * N)
        v0, g0 = m.elbo_and_grad((X, Y))
        tr = training.SVGPTrainer(m, learning_rate=1e-2)
        assert len(tr.host) == (3 if shared else 7)            # distinct kernel Parameters + the noise variance
        before = {n: np.array(p.unconstrained_variable, dtype=np.float64, copy=True) for n, p in tr.host.items()}
        f0 = float(tr.step((X, Y)).cpu()[0])
        assert abs(f0 - v0) <= 1e-9 * abs(v0)
        for n, p in tr.host.items():
            du = tr.u[n] - before[n]
            np.testing.assert_allclose(du, 1e-2 * np.sign(np.asarray(g0[p]).reshape(du.shape)), rtol=0, atol=1e-6, err_msg=n)
        fs = [float(tr.step((X, Y)).cpu()[0]) for _ in range(12)]
        assert fs[-1] > f0
        tr.sync_to_model()
        assert abs(float(m.elbo((X, Y)).cpu()) - float(tr.step((X, Y)).cpu()[0])) <= 1e-8 * abs(fs[-1])
    if not shared:
        # ... and together with a heteroskedastic likelihood: the noise Function's Parameters join the kernel members' on the host
        Xh = rng.random((N, 2)); Yh = np.sin(5 * Xh[:, :1]) + (0.7 - 0.6 * Xh[:, :1]) * rng.standard_normal((N, 1))
        for wh in (True, False):
            kern = (gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.25, 0.9])
                    + gpflow.kernels.Matern32(variance=0.6, lengthscales=0.8, active_dims=[1])) * gpflow.kernels.Matern52(variance=0.9, lengthscales=1.2)
            lik = gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=np.array([[-0.3], [0.05]]), b=np.array([0.6])))
            mh = gpflow.models.SVGP(kern, lik, Xh[:M].copy(), q_mu=q_mu[:, :1], q_sqrt=qs[:1], whiten=wh, num_data=5 * N)
            v0, g0 = mh.elbo_and_grad((Xh, Yh))
            tr = training.SVGPTrainer(mh, learning_rate=1e-2)
            assert len(tr.host) == 8                            # six kernel Parameters + A + b
            before = {n: np.array(p.unconstrained_variable, dtype=np.float64, copy=True) for n, p in tr.host.items()}
            f0 = float(tr.step((Xh, Yh)).cpu()[0])
            assert abs(f0 - v0) <= 1e-7 * abs(v0)
            for n, p in tr.host.items():
                du = tr.u[n] - before[n]
                np.testing.assert_allclose(du, 1e-2 * np.sign(np.asarray(g0[p]).reshape(du.shape)), rtol=0, atol=1e-6, err_msg=n)
            assert float(tr.step((Xh, Yh)).cpu()[0]) > f0 or [float(tr.step((Xh, Yh)).cpu()[0]) for _ in range(8)][-1] > f0


@pytest.mark.parametrize("het", [False, True])
def test_trainer_with_separate_kernels_per_latent(gpu, het):
    """SVGPTrainer with one kernel per latent over SHARED inducing points (BASELINE config C5, separate; conditionals/util.py:566-629: P
    independent single-output problems that share Z, the likelihood and the minibatch rows), with a constant noise variance or a noise
    Function: first objective == SVGP.elbo_and_grad, first Adam step == learning rate along that gradient for every kernel / noise
    parameter and for Z, the bound improves, and the synced model reproduces the trainer's next objective."""
    import gpflow_amd as gpflow
    from gpflow_amd import training
    rng = np.random.default_rng(9)
    N, M, P = 160, 18, 2
    X = rng.random((N, 2)); Y = np.sin(5 * X[:, :1]) + 0.2 * rng.standard_normal((N, P))
    Z = X[:M].copy(); q_mu = 0.2 * rng.normal(size=(M, P)); qs = np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)
    for wh in (True, False):
        ks = [gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.25, 0.9]),
              gpflow.kernels.Matern32(variance=0.7, lengthscales=0.5, active_dims=[0])]
        lik = gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=np.array([[-0.3], [0.05]]), b=np.array([0.6]))) if het \
            else gpflow.likelihoods.Gaussian(0.2)
        iv = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Z.copy()))
        m = gpflow.models.SVGP(gpflow.kernels.SeparateIndependent(ks), lik, iv, q_mu=q_mu, q_sqrt=qs, whiten=wh, num_latent_gps=P, num_data=5 * N)
        v0, g0 = m.elbo_and_grad((X, Y))
        tr = training.SVGPTrainer(m, learning_rate=1e-2)
        assert len(tr.host) == (6 if het else 5)
        before = {n: np.array(p.unconstrained_variable, dtype=np.float64, copy=True) for n, p in tr.host.items()}
        Z0 = tr.dev["Z"].clone()
        f0 = float(tr.step((X, Y)).cpu()[0])
        assert abs(f0 - v0) <= 1e-7 * abs(v0)
        for n, p in tr.host.items():
            du = tr.u[n] - before[n]
            np.testing.assert_allclose(du, 1e-2 * np.sign(np.asarray(g0[p]).reshape(du.shape)), rtol=0, atol=1e-6, err_msg=n)
        np.testing.assert_allclose((tr.dev["Z"] - Z0).cpu().numpy(), 1e-2 * np.sign(np.asarray(g0[iv.inducing_variable.Z])), rtol=0, atol=1e-5)
        fs = [float(tr.step((X, Y)).cpu()[0]) for _ in range(10)]
        assert fs[-1] > f0
        tr.sync_to_model()
        assert abs(float(m.elbo((X, Y)).cpu()) - float(tr.step((X, Y)).cpu()[0])) <= 1e-7 * abs(fs[-1])


def test_heteroskedastic_noise_with_separate_kernels_per_latent(gpu):
    """SeparateIndependent kernels over shared inducing points under Gaussian(scale=Linear(A, b)): the latents share the likelihood, so
    their per-row dF/d sigma_n^2 add up before the noise Function's reverse pass.  Oracle: the sum over the latents of the single-output
    autograd problems (conditionals/util.py:566-629 treats them as independent problems)."""
    import gpflow_amd as gpflow
    rng = np.random.default_rng(9)
    N, M, P = 160, 18, 2
    X = rng.random((N, 2)); Y = np.sin(5 * X[:, :1]) + (0.7 - 0.6 * X[:, :1]) * rng.standard_normal((N, P))
    A0, b0 = np.array([[-0.3], [0.05]]), np.array([0.6])
    Z = X[:M].copy(); q_mu = 0.2 * rng.normal(size=(M, P)); qs = np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)
    hy = [(1.1, [0.25, 0.9]), (0.7, [0.4, 0.6])]
    for wh in (True, False):
        kern = gpflow.kernels.SeparateIndependent([gpflow.kernels.SquaredExponential(variance=v, lengthscales=l_) for v, l_ in hy])
        iv = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Z.copy()))
        m = gpflow.models.SVGP(kern, gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=A0.copy(), b=b0.copy())), iv, q_mu=q_mu,
                               q_sqrt=qs, whiten=wh, num_latent_gps=P, num_data=5 * N)
        v, g = m.elbo_and_grad((X, Y))
        rv, gA, gb, gZ = 0.0, 0.0, 0.0, 0.0
        for p in range(P):
            r, rg = orcg.heteroskedastic_value_and_grads("svgp" if wh else "svgp_unwhitened", X, Y[:, p:p + 1], A=A0, b=b0, variance=hy[p][0],
                                                         lengthscales=hy[p][1], Z=Z, q_mu=q_mu[:, p:p + 1], q_sqrt=qs[p:p + 1], num_data=5 * N)
            rv += r; gA = gA + rg["A"]; gb = gb + rg["b"]; gZ = gZ + rg["Z"]
        assert abs(v - rv) <= 1e-7 * abs(rv) and abs(v - float(m.elbo((X, Y)).cpu())) <= 1e-7 * abs(v)
        for got, ref in ((g[m.likelihood.scale.A], gA), (g[m.likelihood.scale.b], gb), (g[iv.inducing_variable.Z], gZ)):
            got = np.asarray(got, dtype=np.float64).reshape(np.shape(ref))
            assert np.abs(got - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max())


def test_heteroskedastic_noise_under_a_kernel_combination(gpu):
    """Gaussian(scale=Linear(A, b)) together with a NESTED kernel combination, (SquaredExponential + Matern32[dim 1]) * Matern52, through
    the model surface of GPR, both SVGP parametrisations and SGPR: value and the gradients w.r.t. the noise Function's parameters, every
    member's variance and lengthscales and Z against torch autograd over the restated model (1e-7: the un-whitened / SGPR chains)."""
    import gpflow_amd as gpflow
    rng = np.random.default_rng(20220701)
    N, M, P = 200, 24, 1
    X = rng.random((N, 2)); Y = np.sin(5 * X[:, :1]) + (0.7 - 0.6 * X[:, :1]) * rng.standard_normal((N, 1))
    A0, b0 = np.array([[-0.3], [0.05]]), np.array([0.6])
    Z = X[:M] + 0.02 * rng.normal(size=(M, 2)); q_mu = 0.2 * rng.normal(size=(M, P))
    qs = np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)
    members = [("SquaredExponential", 1.1, np.array([0.25, 0.9])), ("Matern32", 0.6, np.array(0.8)), ("Matern52", 0.9, np.array(1.2))]
    combo = (members, ("mul", [("add", [0, 1]), 2]), [None, [1], None])
    mk_lik = lambda: gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=A0.copy(), b=b0.copy()))  # noqa: E731

    def mk_k():
        return (gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.25, 0.9])
                + gpflow.kernels.Matern32(variance=0.6, lengthscales=0.8, active_dims=[1])) * gpflow.kernels.Matern52(variance=0.9, lengthscales=1.2)

    def check(model, v, g, rv, rg, with_z):
        assert abs(v - rv) <= 1e-7 * abs(rv), (v, rv)
        ks = model.kernel.kernels
        leaves = [ks[0].kernels[0], ks[0].kernels[1], ks[1]]
        lik = model.likelihood

        def rel(got, ref):
            got = np.asarray(got, dtype=np.float64).reshape(np.shape(ref))
            return np.abs(got - ref).max() / max(1.0, np.abs(np.asarray(ref)).max())
        assert rel(g[lik.scale.A], rg["A"]) <= 1e-7 and rel(g[lik.scale.b], rg["b"]) <= 1e-7
        for i, k in enumerate(leaves):
            for par, ref in ((k.variance, rg["variance"][i]), (k.lengthscales, rg["lengthscales"][i])):
                got = np.ravel(g[par]) / np.ravel(par.transform.forward_grad(par.unconstrained_variable))
                assert rel(got, np.ravel(ref)) <= 1e-7, (i, par.name)
        if with_z:
            assert rel(g[model.inducing_variable.Z], rg["Z"]) <= 1e-7
    m = gpflow.models.GPR((X, Y), mk_k(), likelihood=mk_lik())
    v, g = m.objective_and_grad()
    check(m, v, g, *orcg.heteroskedastic_value_and_grads("gpr", X, Y, A=A0, b=b0, combination=combo), with_z=False)
    for wh, name in ((True, "svgp"), (False, "svgp_unwhitened")):
        s = gpflow.models.SVGP(mk_k(), mk_lik(), Z.copy(), q_mu=q_mu, q_sqrt=qs, whiten=wh, num_data=5 * N)
        v, g = s.elbo_and_grad((X, Y))
        check(s, v, g, *orcg.heteroskedastic_value_and_grads(name, X, Y, A=A0, b=b0, Z=Z, q_mu=q_mu, q_sqrt=qs, num_data=5 * N,
                                                              combination=combo), with_z=True)
    sg = gpflow.models.SGPR((X, Y), mk_k(), Z.copy(), likelihood=mk_lik())
    v, g = sg.objective_and_grad()
    check(sg, v, g, *orcg.heteroskedastic_value_and_grads("sgpr", X, Y, A=A0, b=b0, Z=Z, combination=combo), with_z=True)


@pytest.mark.parametrize("q_diag", [False, True])
def test_heteroskedastic_noise_in_the_reverse_pass(gpu, q_diag):
    """Gaussian(scale=Linear(A, b)) (the reference's tests/integration/test_linear_noise.py recipe; likelihoods/scalar_continuous.py:
    52-148): GPR.log_marginal_likelihood and the whitened SVGP.elbo with their gradients w.r.t. the NOISE FUNCTION's parameters (A, b)
    and everything else -- dF/d sigma_n^2 per row from the device reverse pass, chained through the clip and the function's own
    reverse pass -- against torch autograd over the restated likelihood; Scipy then fits the noise slope."""
    import gpflow_amd as gpflow
    rng = np.random.default_rng(20220630)
    N, M, P = 240, 30, 1
    X = rng.random((N, 2)); Xc = X[:, :1]
    Y = np.sin(5 * Xc) + (0.7 - 0.6 * Xc) * rng.standard_normal((N, 1))
    A0, b0 = np.array([[-0.3], [0.05]]), np.array([0.6])
    Z = X[:M] + 0.01 * rng.normal(size=(M, 2)); q_mu = 0.2 * rng.normal(size=(M, P))
    qs = 0.4 + np.abs(rng.normal(size=(M, P))) if q_diag else np.tril(0.1 * rng.normal(size=(P, M, M))) + 0.5 * np.eye(M)
    mk_lik = lambda: gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=A0.copy(), b=b0.copy()))  # noqa: E731
    mk_k = lambda: gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.25, 0.9])  # noqa: E731

    def chk(got, ref, tol=1e-8):
        got = np.asarray(got, dtype=np.float64).reshape(np.shape(ref))
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (np.abs(got - ref).max(), np.abs(ref).max())

    def unc(m, par, g):   # d/d(constrained) from the model's d/d(unconstrained)
        return np.asarray(g[par]) / par.transform.forward_grad(par.unconstrained_variable)
    if not q_diag:
        m = gpflow.models.GPR((X, Y), mk_k(), likelihood=mk_lik())
        v, g = m.objective_and_grad()
        rv, rg = orcg.heteroskedastic_value_and_grads("gpr", X, Y, A=A0, b=b0, variance=1.1, lengthscales=[0.25, 0.9])
        assert abs(v - rv) <= 1e-9 * abs(rv) and abs(v - float(m.log_marginal_likelihood().cpu())) <= 1e-9 * abs(v)
        chk(g[m.likelihood.scale.A], rg["A"]); chk(g[m.likelihood.scale.b], rg["b"])
        chk(unc(m, m.kernel.variance, g), rg["variance"]); chk(unc(m, m.kernel.lengthscales, g), rg["lengthscales"])
        before = v
        res = gpflow.optimizers.Scipy().minimize(m, options=dict(maxiter=60))
        assert -res.fun > before + 5.0
        assert abs(float(np.ravel(m.likelihood.scale.A.numpy())[0]) + 0.6) < 0.25      # the slope the data were drawn with
    s = gpflow.models.SVGP(mk_k(), mk_lik(), Z.copy(), q_mu=q_mu, q_sqrt=qs, q_diag=q_diag, num_data=5 * N)
    v, g = s.elbo_and_grad((X, Y))
    rv, rg = orcg.heteroskedastic_value_and_grads("svgp", X, Y, A=A0, b=b0, variance=1.1, lengthscales=[0.25, 0.9], Z=Z, q_mu=q_mu,
                                                  q_sqrt=qs, num_data=5 * N)
    assert abs(v - rv) <= 1e-9 * abs(rv) and abs(v - float(s.elbo((X, Y)).cpu())) <= 1e-9 * abs(v)
    chk(g[s.likelihood.scale.A], rg["A"]); chk(g[s.likelihood.scale.b], rg["b"])
    chk(g[s.inducing_variable.Z], rg["Z"]); chk(g[s.q_mu], rg["q_mu"])
    chk(unc(s, s.kernel.variance, g), rg["variance"]); chk(unc(s, s.kernel.lengthscales, g), rg["lengthscales"])
    if q_diag:
        chk(unc(s, s.q_sqrt, g), rg["q_sqrt"])
    # the un-whitened SVGP under the same likelihood (refused until the end of round 5): the same per-row dF/d sigma_n^2 through the
    # A2t = At Lm^-1 chain; 1e-7 like every un-whitened gradient (one more kappa(Lm))
    su = gpflow.models.SVGP(mk_k(), mk_lik(), Z.copy(), q_mu=q_mu, q_sqrt=qs, q_diag=q_diag, whiten=False, num_data=5 * N)
    v, g = su.elbo_and_grad((X, Y))
    rv, rg = orcg.heteroskedastic_value_and_grads("svgp_unwhitened", X, Y, A=A0, b=b0, variance=1.1, lengthscales=[0.25, 0.9], Z=Z,
                                                  q_mu=q_mu, q_sqrt=qs, num_data=5 * N)
    # (value at 1e-7 too: this ELBO of -5.6e6 is dominated by the KL term through Kuu^-1 of nearly coincident inducing points; LAPACK
    #  and the HIP factorisation differ by 1.0e-9 of it)
    assert abs(v - rv) <= 1e-7 * abs(rv) and abs(v - float(su.elbo((X, Y)).cpu())) <= 1e-7 * abs(v)
    chk(g[su.likelihood.scale.A], rg["A"], 1e-7); chk(g[su.likelihood.scale.b], rg["b"], 1e-7)
    chk(g[su.inducing_variable.Z], rg["Z"], 1e-7); chk(g[su.q_mu], rg["q_mu"], 1e-7)
    chk(unc(su, su.kernel.variance, g), rg["variance"], 1e-7); chk(unc(su, su.kernel.lengthscales, g), rg["lengthscales"], 1e-7)
    if q_diag:
        chk(unc(su, su.q_sqrt, g), rg["q_sqrt"], 1e-7)
    else:
        # SGPR under the same likelihood (sgpr.py:207-211 with one sigma_n per row): dF/d sigma_n^2 per row from the reverse pass
        sgm = gpflow.models.SGPR((X, Y), mk_k(), Z.copy(), likelihood=mk_lik())
        v, g = sgm.objective_and_grad()
        rv, rg = orcg.heteroskedastic_value_and_grads("sgpr", X, Y, A=A0, b=b0, variance=1.1, lengthscales=[0.25, 0.9], Z=Z)
        assert abs(v - rv) <= 1e-9 * abs(rv) and abs(v - float(sgm.elbo().cpu())) <= 1e-9 * abs(v)
        chk(g[sgm.likelihood.scale.A], rg["A"], 1e-7); chk(g[sgm.likelihood.scale.b], rg["b"], 1e-7)
        chk(g[sgm.inducing_variable.Z], rg["Z"], 1e-7)
        chk(unc(sgm, sgm.kernel.variance, g), rg["variance"], 1e-7); chk(unc(sgm, sgm.kernel.lengthscales, g), rg["lengthscales"], 1e-7)
        res = gpflow.optimizers.Scipy().minimize(sgm, options=dict(maxiter=40))
        assert -res.fun > v + 1.0
    # the device-resident trainer under the same likelihood: the noise Function's Parameters are host-side hyper-parameters, sigma_n^2 at
    # the minibatch rows is formed on the device every step.  Its first objective is elbo_and_grad's, its first Adam step moves every
    # noise parameter by the learning rate along the gradient (m / sqrt(v) = sign(g) at t = 1), and it improves the bound.
    from gpflow_amd import training
    for wh in (True, False):
        mt = gpflow.models.SVGP(mk_k(), mk_lik(), Z.copy(), q_mu=q_mu, q_sqrt=qs, q_diag=q_diag, whiten=wh, num_data=5 * N)
        v0, g0 = mt.elbo_and_grad((X, Y))
        pars = list(mt.likelihood.scale.parameters)
        u_before = [np.array(p.unconstrained_variable, dtype=np.float64, copy=True) for p in pars]
        tr = training.SVGPTrainer(mt, learning_rate=1e-2)
        f0 = float(tr.step((X, Y)).cpu()[0])
        assert abs(f0 - v0) <= 1e-9 * abs(v0)
        for i, p in enumerate(pars):
            du = tr.u[f"noise_fn_{i}"] - u_before[i]
            np.testing.assert_allclose(du, 1e-2 * np.sign(np.asarray(g0[p]).reshape(du.shape)), rtol=0, atol=1e-6)
        fs = [float(tr.step((X, Y)).cpu()[0]) for _ in range(20)]
        assert fs[-1] > f0
        tr.sync_to_model()
        assert abs(float(mt.elbo((X, Y)).cpu()) - float(tr.step((X, Y)).cpu()[0])) <= 1e-8 * abs(fs[-1])


@pytest.mark.parametrize("rows", [1, 2])
def test_heteroskedastic_noise_on_a_one_row_minibatch(gpu, rows):
    """A per-row noise vector with ONE entry is still a per-row vector (the last minibatch when N mod B == 1): the reverse passes
    used to take the constant-noise branch for it and fail on float(np.log(tensor)).  SVGP whitened / un-whitened and GPR on 1 (and,
    as the control, 2) rows against torch autograd over the restated likelihood (likelihoods/scalar_continuous.py:92-148)."""
    import gpflow_amd as gpflow
    rng = np.random.default_rng(7)
    M = 12
    X = rng.random((rows, 2)); Y = rng.standard_normal((rows, 1))
    A0, b0 = np.array([[-0.3], [0.05]]), np.array([0.6])
    Z = rng.random((M, 2)); q_mu = 0.2 * rng.normal(size=(M, 1))
    qs = np.tril(0.1 * rng.normal(size=(1, M, M))) + 0.5 * np.eye(M)
    mk_lik = lambda: gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=A0.copy(), b=b0.copy()))  # noqa: E731
    mk_k = lambda: gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.25, 0.9])  # noqa: E731

    def chk(got, ref, tol=1e-8):
        got = np.asarray(got, dtype=np.float64).reshape(np.shape(ref))
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (np.abs(got - ref).max(), np.abs(ref).max())
    for whiten, name, tol in ((True, "svgp", 1e-8), (False, "svgp_unwhitened", 1e-7)):
        s = gpflow.models.SVGP(mk_k(), mk_lik(), Z.copy(), q_mu=q_mu, q_sqrt=qs, whiten=whiten, num_data=50)
        v, g = s.elbo_and_grad((X, Y))
        rv, rg = orcg.heteroskedastic_value_and_grads(name, X, Y, A=A0, b=b0, variance=1.1, lengthscales=[0.25, 0.9], Z=Z, q_mu=q_mu,
                                                      q_sqrt=qs, num_data=50)
        assert abs(v - rv) <= tol * abs(rv)
        chk(g[s.likelihood.scale.A], rg["A"], tol); chk(g[s.likelihood.scale.b], rg["b"], tol); chk(g[s.q_mu], rg["q_mu"], tol)
    m = gpflow.models.GPR((X, Y), mk_k(), likelihood=mk_lik())
    v, g = m.objective_and_grad()
    rv, rg = orcg.heteroskedastic_value_and_grads("gpr", X, Y, A=A0, b=b0, variance=1.1, lengthscales=[0.25, 0.9])
    assert abs(v - rv) <= 1e-9 * max(1.0, abs(rv))
    chk(g[m.likelihood.scale.A], rg["A"]); chk(g[m.likelihood.scale.b], rg["b"])
