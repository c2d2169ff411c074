"""GPU parity of the gpflow_amd surface against values computed by the REFERENCE'S OWN SOURCE
(tests/golden/ref_golden.npz, written by tests/golden/make_golden_ref.py: the unmodified GPflow package run over NumPy
stand-ins for TensorFlow).  Every call goes Python host -> ctypes -> libgpk.so.  The same bodies run in the CPU tier
against the emulated primitives (tests/test_host_emulated.py), which checks the host layer's routing and shapes.

Tolerances: 1e-9 absolute on O(1) quantities (two correct fp64 algorithms: the reference solves triangular systems by
substitution, the device multiplies by explicit inverses of 128-blocks), 1e-9 relative on ELBO / LML scalars; the cached
posteriors go through Kuu^-1-like products and get 1e-7.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_golden.npz"))


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def close(a, b, atol=1e-9):
    a, b = _np(a), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    assert float(np.max(np.abs(a - b))) <= atol * scale, float(np.max(np.abs(a - b))) / scale


@pytest.fixture(scope="module")
def gp(gpu):
    import gpflow_amd
    return gpflow_amd


def test_ref_kernels(gp):
    X, X2, ls, var = G["k_X"], G["k_X2"], G["k_ls"], float(G["k_var"])
    for name in ("SquaredExponential", "Matern12", "Matern32", "Matern52"):
        k = getattr(gp.kernels, name)(variance=var, lengthscales=ls)
        # Matern K(X, X): r = sqrt(max(r2, 1e-36)) with r2 from the expansion formula, which is 0 +- a few ulp of |x|^2 on
        # the diagonal -- r, and with it exp(-r), carries sqrt(ulp) ~ 1e-8 of rounding noise THERE in the reference itself
        # (kernels/stationaries.py:111-116); any other summation order lands elsewhere inside that noise.
        sym_tol = 1e-12 if name == "SquaredExponential" else 3e-7
        close(k(X), G[f"k_{name}_sym"], sym_tol); close(k(X, X2), G[f"k_{name}_cross"], 1e-12)
        off = ~np.eye(len(X), dtype=bool)
        assert np.max(np.abs(_np(k(X)) - G[f"k_{name}_sym"])[off]) <= 1e-12 * var
        close(k(X, full_cov=False), G[f"k_{name}_diag"], 1e-14)
    k_ad = gp.kernels.SquaredExponential(variance=0.8, lengthscales=[0.5, 1.5], active_dims=[1, 3])
    k_sl = gp.kernels.SquaredExponential(variance=0.8, lengthscales=0.6, active_dims=slice(0, 2))
    close(k_ad(X), G["k_active_dims_sym"], 1e-12)
    close((k_ad + k_sl)(X, X2), G["k_sum_cross"], 1e-12)
    close((k_ad * gp.kernels.Matern32(variance=1.1, lengthscales=0.9))(X), G["k_prod_sym"], 3e-7)  # (Matern diagonal, see above)
    close((k_ad + k_sl)(X, full_cov=False), G["k_sum_diag"], 1e-14)


def test_ref_heteroskedastic_gaussian(gp):
    """Gaussian(scale=Linear(A, b)) and Gaussian(variance=Polynomial) (likelihoods/scalar_continuous.py:52-148): per-row noise variances
    through gpk_gpr_lml / gpk_diag_add (add_likelihood_noise_cov, utilities/model_utils.py:46-50) and the per-row variational
    expectations inside gpk_svgp_elbo_shard (whitened, un-whitened, q_diag), against the values of the reference's own source."""
    X, Y, Xs = G["het_X"], G["het_Y"], G["het_Xnew"]
    mk_lik = lambda: gp.likelihoods.Gaussian(scale=gp.functions.Linear(A=G["het_A"], b=G["het_b"]))  # noqa: E731
    mk_k = lambda: gp.kernels.SquaredExponential(variance=1.3, lengthscales=[0.3, 0.6])  # noqa: E731
    lik = mk_lik()
    assert lik.is_heteroskedastic
    close(lik.variance_at(X), G["het_variance_at"], 1e-14)
    m = gp.models.GPR((X, Y), mk_k(), likelihood=lik)
    np.testing.assert_allclose(float(m.log_marginal_likelihood()), float(G["het_gpr_lml"]), rtol=1e-10)
    mu, var = m.predict_f(Xs)
    close(mu, G["het_gpr_fmu"]); close(var, G["het_gpr_fvar"])
    ymu, yvar = m.predict_y(Xs)
    close(ymu, G["het_gpr_ymu"]); close(yvar, G["het_gpr_yvar"])
    close(m.predict_log_density((Xs, np.cos(Xs[:, :1]))), G["het_gpr_logdens"])
    pmu, pvar = m.posterior().predict_f(Xs)
    close(pmu, G["het_gpr_cached_mu"], 1e-7); close(pvar, G["het_gpr_cached_var"], 1e-7)
    # a non-stationary route to the same LML: the composed path (K + diag(noise) by gpk_diag_add, then the primitives)
    ksum = mk_k() + gp.kernels.SquaredExponential(variance=1e-12, lengthscales=1.0)
    np.testing.assert_allclose(float(gp.models.GPR((X, Y), ksum, likelihood=mk_lik()).log_marginal_likelihood()),
                               float(G["het_gpr_lml"]), rtol=1e-8)
    for wh, name in ((True, "white"), (False, "unwhite")):
        s = gp.models.SVGP(mk_k(), mk_lik(), G["het_Z"], q_mu=G["het_q_mu"], q_sqrt=G["het_q_sqrt"], whiten=wh, num_data=400)
        np.testing.assert_allclose(float(s.elbo((X, Y))), float(G[f"het_svgp_elbo_{name}"]), rtol=1e-9)
        if wh:   # predictions through the likelihood: the noise Function evaluated at the NEW inputs
            symu, syvar = s.predict_y(Xs)
            close(symu, G["het_svgp_ymu"], 1e-8); close(syvar, G["het_svgp_yvar"], 1e-8)
            close(s.predict_log_density((Xs, np.cos(Xs[:, :1]))), G["het_svgp_logdens"], 1e-8)
    likp = lambda: gp.likelihoods.Gaussian(variance=gp.functions.Polynomial(2, input_dim=2, w=G["het_poly_w"]),  # noqa: E731
                                           variance_lower_bound=1e-3)
    close(likp().variance_at(X), G["het_poly_variance_at"], 1e-14)
    np.testing.assert_allclose(float(gp.models.GPR((X, Y), mk_k(), likelihood=likp()).log_marginal_likelihood()),
                               float(G["het_poly_gpr_lml"]), rtol=1e-9)
    sq = gp.models.SVGP(mk_k(), likp(), G["het_Z"], q_mu=G["het_q_mu"], q_sqrt=G["het_poly_q_sqrt_diag"], q_diag=True, num_data=400)
    np.testing.assert_allclose(float(sq.elbo((X, Y))), float(G["het_poly_svgp_elbo_qdiag"]), rtol=1e-9)
    # SGPR under the same likelihood: rows of A^T and err scaled by 1 / sigma_n (sgpr.py:207-211), two more scalars in the statistics
    sg = gp.models.SGPR((X, Y), mk_k(), G["het_Z"], likelihood=mk_lik())
    np.testing.assert_allclose(float(sg.elbo()), float(G["het_sgpr_elbo"]), rtol=1e-9)
    smu, svar = sg.predict_f(Xs)
    close(smu, G["het_sgpr_mu"], 1e-8); close(svar, G["het_sgpr_var"], 1e-8)
    qmu, qcov = sg.compute_qu()
    close(qmu, G["het_sgpr_qu_mu"], 1e-7); close(qcov, G["het_sgpr_qu_cov"], 1e-7)
    gymu, gyvar = sg.predict_y(Xs)
    close(gymu, G["het_sgpr_ymu"], 1e-8); close(gyvar, G["het_sgpr_yvar"], 1e-8)
    close(sg.predict_log_density((Xs, np.cos(Xs[:, :1]))), G["het_sgpr_logdens"], 1e-8)
    # the upper bound rescales every row by its own sigma_n^2 + c (sgpr.py:124-131): a second statistics pass
    np.testing.assert_allclose(float(sg.upper_bound()), float(G["het_sgpr_upper"]), rtol=1e-9)
    vs, gs = sg.objective_and_grad()          # (gradients through the noise function under SGPR: end of round 5)
    np.testing.assert_allclose(vs, float(G["het_sgpr_elbo"]), rtol=1e-9)
    assert sg.likelihood.scale.A in gs and sg.inducing_variable.Z in gs
    # gradients through the noise function: GPR and both SVGP parametrisations have them (tests/test_gpu_gradients.py); here the
    # un-whitened model's value from the reverse pass agrees with the reference's ELBO
    assert m.likelihood.scale.A in m.log_marginal_likelihood_and_grad()[1]
    vu, gu = s.elbo_and_grad((X, Y))
    np.testing.assert_allclose(vu, float(G["het_svgp_elbo_unwhite"]), rtol=1e-9)
    assert s.likelihood.scale.A in gu and s.likelihood.scale.b in gu


def test_ref_gpr(gp):
    m = gp.models.GPR((G["gpr_X"], G["gpr_Y"]), gp.kernels.SquaredExponential(variance=1.0, lengthscales=2.0), noise_variance=1.0)
    Xn = G["gpr_Xnew"]
    np.testing.assert_allclose(float(m.log_marginal_likelihood()), float(G["gpr_lml"]), rtol=1e-10)
    mu, var = m.predict_f(Xn)
    close(mu, G["gpr_mu"]); close(var, G["gpr_var"])
    close(m.predict_f(Xn, full_cov=True)[1], G["gpr_var_fullcov"])
    ymu, yvar = m.predict_y(Xn)
    close(ymu, G["gpr_ymu"]); close(yvar, G["gpr_yvar"])
    close(m.predict_log_density((Xn, np.cos(Xn))), G["gpr_logdens"])
    pmu, pvar = m.posterior().predict_f(Xn)
    close(pmu, G["gpr_cached_mu"]); close(pvar, G["gpr_cached_var"])
    # two output columns, constant mean, ARD, Xnew with leading batch dims -- incl. full_cov over the batch dims
    m2 = gp.models.GPR((G["gpr2_X"], G["gpr2_Y"]), gp.kernels.SquaredExponential(variance=1.3, lengthscales=G["gpr2_ls"]),
                       mean_function=gp.mean_functions.Constant(np.array([0.3])), noise_variance=0.07)
    np.testing.assert_allclose(float(m2.log_marginal_likelihood()), float(G["gpr2_lml"]), rtol=1e-10)
    Xb = G["gpr2_Xnew"]
    mu, var = m2.predict_f(Xb)
    close(mu, G["gpr2_mu"]); close(var, G["gpr2_var"])
    mu_fc, var_fc = m2.predict_f(Xb, full_cov=True)
    close(mu_fc, G["gpr2_mu"]); close(var_fc, G["gpr2_var_fullcov"])
    pmu, pvar = m2.posterior().predict_f(Xb, full_cov=True)
    close(pmu, G["gpr2_mu"]); close(pvar, G["gpr2_var_fullcov"])
    # BASELINE config C1
    m1 = gp.models.GPR((G["c1_X"], G["c1_Y"]), gp.kernels.SquaredExponential(), noise_variance=0.1)
    np.testing.assert_allclose(float(m1.log_marginal_likelihood()), float(G["c1_lml"]), rtol=1e-10)
    mu, var = m1.predict_f(G["c1_Xnew"])
    close(mu, G["c1_mu"]); close(var, G["c1_var"])


def test_ref_gauss_kl(gp):
    kl = gp.kullback_leiblers.gauss_kl
    mu, sq, K, Kb = G["kl_mu"], G["kl_sqrt"], G["kl_K"], G["kl_Kb"]
    for val, key in ((kl(mu, sq), "kl_white"), (kl(mu, sq, K), "kl_K_val"), (kl(mu, sq, Kb), "kl_Kb_val"),
                     (kl(mu, G["kl_sqrt_diag"]), "kl_diag_white"), (kl(mu, G["kl_sqrt_diag"], K), "kl_diag_K"),
                     (kl(mu, G["kl_sqrt_upper"], K), "kl_upper_ignored"),
                     (kl(mu, sq, K_cholesky=np.linalg.cholesky(K)), "kl_Kchol")):
        np.testing.assert_allclose(float(val), float(G[key]), rtol=1e-8)  # (K has a 1e-6 jitter: kappa ~ 1e7)


def test_ref_conditionals(gp):
    Z, Xn, f, qs, qd = G["cond_Z"], G["cond_X"], G["cond_f"], G["cond_qs"], G["cond_qd"]
    kern = gp.kernels.SquaredExponential(variance=1.4, lengthscales=[0.8, 1.2])
    iv = gp.inducing_variables.InducingPoints(Z)
    for white in (False, True):
        for fc in (False, True):
            for tag, q in (("full", qs), ("diag", qd), ("none", None)):
                mu, var = gp.conditionals.conditional(Xn, iv, kern, f, full_cov=fc, q_sqrt=q, white=white)
                close(mu, G[f"cond_w{int(white)}_fc{int(fc)}_{tag}_mu"], 1e-8); close(var, G[f"cond_w{int(white)}_fc{int(fc)}_{tag}_var"], 1e-8)
    Xb = G["cond_Xb"]  # [2, 3, N, D]: leading batch dims, with and without full_cov (util.py:108-131)
    for white, key in ((True, "cond_batch"), (False, "cond_batch_unw")):
        for fc in (False, True):
            mu, var = gp.conditionals.conditional(Xb, iv, kern, f, full_cov=fc, q_sqrt=qs, white=white)
            close(mu, G[f"{key}_fc{int(fc)}_mu"], 1e-8); close(var, G[f"{key}_fc{int(fc)}_var"], 1e-8)
    mu, var = gp.conditionals.base_conditional(G["bc_Kmn"], G["bc_Kmm"], G["bc_Knn"], f, q_sqrt=qs, white=False)
    close(mu, G["bc_mu"], 1e-8); close(var, G["bc_var"], 1e-8)


def test_ref_svgp(gp):
    X, Y, Z, q_mu, q_sqrt, Xs = (G[k] for k in ("svgp_X", "svgp_Y", "svgp_Z", "svgp_q_mu", "svgp_q_sqrt", "svgp_Xnew"))
    for w in (0, 1):
        s = gp.models.SVGP(gp.kernels.SquaredExponential(variance=1.0, lengthscales=1.0), gp.likelihoods.Gaussian(variance=1.0),
                           Z.copy(), q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), whiten=bool(w), num_latent_gps=2)
        np.testing.assert_allclose(float(s.elbo((X, Y))), float(G[f"svgp_elbo_w{w}"]), rtol=1e-9)
        np.testing.assert_allclose(float(s.prior_kl()), float(G[f"svgp_kl_w{w}"]), rtol=1e-9)
        mu, var = s.predict_f(Xs)
        close(mu, G[f"svgp_mu_w{w}"]); close(var, G[f"svgp_var_w{w}"])
        close(s.predict_f(Xs, full_cov=True)[1], G[f"svgp_var_fullcov_w{w}"])
        pmu, pvar = s.posterior().predict_f(Xs)
        close(pmu, G[f"svgp_cached_mu_w{w}"], 1e-7); close(pvar, G[f"svgp_cached_var_w{w}"], 1e-7)
        sd = gp.models.SVGP(gp.kernels.SquaredExponential(variance=1.0, lengthscales=1.0), gp.likelihoods.Gaussian(variance=1.0),
                            Z.copy(), q_mu=q_mu.copy(), q_sqrt=G["svgp_q_sqrt_diag"].copy(), q_diag=True, whiten=bool(w),
                            num_latent_gps=2, num_data=100)
        np.testing.assert_allclose(float(sd.elbo((X, Y))), float(G[f"svgp_elbo_diag_w{w}"]), rtol=1e-9)
        sm = gp.models.SVGP(gp.kernels.SquaredExponential(variance=1.0, lengthscales=G["mid_ls"]), gp.likelihoods.Gaussian(variance=0.1),
                            G["mid_Z"].copy(), q_mu=G["mid_q_mu"].copy(), q_sqrt=G["mid_q_sqrt"].copy(), whiten=bool(w),
                            num_data=100000, mean_function=gp.mean_functions.Constant(np.array([0.2])))
        np.testing.assert_allclose(float(sm.elbo((G["mid_X"], G["mid_Y"]))), float(G[f"mid_elbo_w{w}"]), rtol=1e-9)


def test_ref_multi_output(gp):
    X, Y, Z, Xs, q_mu, q_sqrt = (G[k] for k in ("mo_X", "mo_Y", "mo_Z", "mo_Xnew", "mo_q_mu", "mo_q_sqrt"))
    L = q_mu.shape[1]
    mo = gp.kernels.SharedIndependent(gp.kernels.SquaredExponential(variance=1.2, lengthscales=[0.9, 1.1]), output_dim=L)
    ivs = gp.inducing_variables.SharedIndependentInducingVariables(gp.inducing_variables.InducingPoints(Z.copy()))
    for w in (0, 1):
        s = gp.models.SVGP(mo, gp.likelihoods.Gaussian(variance=0.2), ivs, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), whiten=bool(w),
                           num_latent_gps=L)
        np.testing.assert_allclose(float(s.elbo((X, Y))), float(G[f"mo_shared_elbo_w{w}"]), rtol=1e-9)
        mu, var = s.predict_f(Xs)
        close(mu, G[f"mo_shared_mu_w{w}"]); close(var, G[f"mo_shared_var_w{w}"])
        pmu, pvar = s.posterior().predict_f(Xs)
        close(pmu, G[f"mo_shared_cached_mu_w{w}"], 1e-7); close(pvar, G[f"mo_shared_cached_var_w{w}"], 1e-7)
    lss, vars_, Zs = G["mo_sep_ls"], G["mo_sep_var"], G["mo_sep_Z"]
    ksep = gp.kernels.SeparateIndependent([gp.kernels.SquaredExponential(variance=float(vars_[i]), lengthscales=lss[i]) for i in range(L)])
    ivsep = gp.inducing_variables.SeparateIndependentInducingVariables([gp.inducing_variables.InducingPoints(z.copy()) for z in Zs])
    for w in (0, 1):
        s = gp.models.SVGP(ksep, gp.likelihoods.Gaussian(variance=0.2), ivsep, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), whiten=bool(w),
                           num_latent_gps=L)
        np.testing.assert_allclose(float(s.elbo((X, Y))), float(G[f"mo_sep_elbo_w{w}"]), rtol=1e-9)
        post = s.posterior()  # separate-kernel cache: alpha [L,M,1], Qinv [L,M,M] (posteriors.py:694-746)
        for fc in (0, 1):
            mu, var = s.predict_f(Xs, full_cov=bool(fc))
            close(mu, G[f"mo_sep_mu_w{w}_fc{fc}"]); close(var, G[f"mo_sep_var_w{w}_fc{fc}"])
            pmu, pvar = post.predict_f(Xs, full_cov=bool(fc))
            close(pmu, G[f"mo_sep_cached_mu_w{w}_fc{fc}"], 1e-7); close(pvar, G[f"mo_sep_cached_var_w{w}_fc{fc}"], 1e-7)
        close(s.predict_f(Xs, full_output_cov=True)[1], G[f"mo_sep_var_w{w}_foc"])
        close(post.predict_f(Xs, full_output_cov=True)[1], G[f"mo_sep_var_w{w}_foc"], 1e-7)
    # shared kernel + separate inducing variables: Kff has no latent axis while Kfu does (the cached
    # route indexed Kff by latent and returned a wrong full covariance)
    for w in (0, 1):
        s = gp.models.SVGP(mo, gp.likelihoods.Gaussian(variance=0.2), ivsep, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), whiten=bool(w),
                           num_latent_gps=L)
        np.testing.assert_allclose(float(s.elbo((X, Y))), float(G[f"mo_shsep_elbo_w{w}"]), rtol=1e-9)
        post = s.posterior()
        for fc in (0, 1):
            mu, var = s.predict_f(Xs, full_cov=bool(fc))
            close(mu, G[f"mo_shsep_mu_w{w}_fc{fc}"]); close(var, G[f"mo_shsep_var_w{w}_fc{fc}"])
            pmu, pvar = post.predict_f(Xs, full_cov=bool(fc))
            close(pmu, G[f"mo_shsep_cached_mu_w{w}_fc{fc}"], 1e-7); close(pvar, G[f"mo_shsep_cached_var_w{w}_fc{fc}"], 1e-7)


def test_ref_sgpr(gp):
    sg = gp.models.SGPR((G["sgpr_X"], G["sgpr_Y"]), gp.kernels.SquaredExponential(variance=1.1, lengthscales=[0.8, 1.2]),
                        G["sgpr_Z"].copy(), noise_variance=0.05)
    np.testing.assert_allclose(float(sg.elbo()), float(G["sgpr_elbo"]), rtol=1e-9)
    np.testing.assert_allclose(float(sg.upper_bound()), float(G["sgpr_upper"]), rtol=1e-9)
    mu, var = sg.predict_f(G["sgpr_Xnew"])
    close(mu, G["sgpr_mu"], 1e-8); close(var, G["sgpr_var"], 1e-8)
    qmu, qcov = sg.compute_qu()
    close(qmu, G["sgpr_qu_mu"], 1e-7); close(qcov, G["sgpr_qu_cov"], 1e-7)


def test_ref_gradients(gp):
    """The hand-written reverse pass (gpflow_amd/gradients.py, every product a libgpk call) against gradients of the
    reference's own SVGP.elbo (whitened and not) / GPR.log_marginal_likelihood / SGPR.elbo obtained by Richardson central
    differences of the unmodified GPflow source (make_golden_ref.py).  1e-7 of each gradient's largest entry."""
    from gpflow_amd import gradients, ops
    X, Y, Z, qm, qs = (ops.to_device(G[k]) for k in ("g_X", "g_Y", "g_Z", "g_q_mu", "g_q_sqrt"))
    kw = dict(variance=float(G["g_variance"]), lengthscales=G["g_lengthscales"], noise_variance=float(G["g_noise_variance"]))
    B = X.shape[0]

    def chk(g, ref, tol=1e-7):
        g = _np(g).reshape(ref.shape)
        assert np.abs(g - ref).max() <= tol * max(np.abs(ref).max(), 1e-300), (np.abs(g - ref).max(), np.abs(ref).max())
    for w, fn in ((1, gradients.svgp_elbo_and_grad), (0, gradients.svgp_elbo_and_grad_unwhitened)):
        F, g, info = fn(Z, X, Y, qm, qs, jitter=1e-6, scale=500.0 / B, **kw)
        assert int(_np(info).ravel()[0]) == 0
        np.testing.assert_allclose(float(_np(F).ravel()[0]), float(G[f"g_svgp_elbo_w{w}"]), rtol=1e-10)
        for n in ("variance", "lengthscales", "noise_variance", "Z", "q_mu"):
            chk(g[n], G[f"g_svgp_d{n}_w{w}"])
        chk(np.tril(_np(g["q_sqrt"])), G[f"g_svgp_dq_sqrt_w{w}"])
    Y1 = Y[:, :1].contiguous()
    F, g, info = gradients.gpr_lml_and_grad(X, Y1, **kw)
    np.testing.assert_allclose(float(_np(F).ravel()[0]), float(G["g_gpr_lml"]), rtol=1e-10)
    for n in ("variance", "lengthscales", "noise_variance"):
        chk(g[n], G[f"g_gpr_d{n}"])
    F, g, info = gradients.sgpr_elbo_and_grad(Z, X, Y1, jitter=1e-6, **kw)
    np.testing.assert_allclose(float(_np(F).ravel()[0]), float(G["g_sgpr_elbo"]), rtol=1e-10)
    for n in ("variance", "lengthscales", "noise_variance", "Z"):
        chk(g[n], G[f"g_sgpr_d{n}"])



def test_ref_nested_kernel_combination(gp):
    """(SquaredExponential + Matern32[dim 1]) * Matern52 -- a Product holding a Sum (kernels/base.py:223-329) -- through the model
    surface: GPR.log_marginal_likelihood, the whitened SVGP.elbo, and their gradients (the reverse pass walks the combination tree:
    gradients.KernelSpec with a tree) against the reference's values and Richardson differences of its forward code."""
    def kern():
        return (gp.kernels.SquaredExponential(variance=float(G["nest_v0"]), lengthscales=G["nest_ls0"])
                + gp.kernels.Matern32(variance=float(G["nest_v1"]), lengthscales=0.8, active_dims=[1])) \
            * gp.kernels.Matern52(variance=float(G["nest_v2"]), lengthscales=float(G["nest_ls2"]))
    X, Y = G["g_X"], G["g_Y"]
    m = gp.models.GPR((X, Y[:, :1]), kern(), noise_variance=0.15)
    np.testing.assert_allclose(float(m.log_marginal_likelihood()), float(G["nest_gpr_lml"]), rtol=1e-9)
    s = gp.models.SVGP(kern(), gp.likelihoods.Gaussian(0.15), G["g_Z"].copy(), q_mu=G["g_q_mu"], q_sqrt=G["g_q_sqrt"], num_latent_gps=2,
                       num_data=500)
    np.testing.assert_allclose(float(s.elbo((X, Y))), float(G["nest_svgp_elbo"]), rtol=1e-9)
    for mdl, pre, call in ((m, "nest_gpr", lambda: m.objective_and_grad()), (s, "nest_svgp", lambda: s.elbo_and_grad((X, Y)))):
        v, g = call()
        np.testing.assert_allclose(v, float(G[f"{pre}_lml" if pre == "nest_gpr" else f"{pre}_elbo"]), rtol=1e-9)
        ks = mdl.kernel.kernels            # [Sum(SE, M32), M52]
        leaves = [ks[0].kernels[0], ks[0].kernels[1], ks[1]]
        for i, k in enumerate(leaves):     # d/d(constrained) = d/d(unconstrained) / forward_grad
            par = k.variance
            got = float(np.ravel(g[par])[0] / np.ravel(par.transform.forward_grad(par.unconstrained_variable))[0])
            ref = float(G[f"{pre}_dv{i}"])
            assert abs(got - ref) <= 1e-6 * max(1.0, abs(ref)), (pre, i, got, ref)
        for i, nm in ((0, "ls0"), (2, "ls2")):
            par = leaves[i].lengthscales
            got = np.ravel(g[par]) / np.ravel(par.transform.forward_grad(par.unconstrained_variable))
            ref = np.atleast_1d(G[f"{pre}_d{nm}"])
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (pre, nm)


def test_ref_map_objective_with_priors(gp):
    """MAP objective against the reference's own statements (gpflow/base.py:201-224, models/model.py:47-76 run through the
    shim): log_prior_density with a prior on the constrained value (Gamma, LogNormal) and one on the unconstrained value
    (Normal + log|Jacobian| of the transform), log_posterior_density, training_loss, and the gradient of the log posterior
    w.r.t. the unconstrained variables (reference side: Richardson differences of its forward code)."""
    from gpflow_amd import priors
    from gpflow_amd.base import PriorOn
    m = gp.models.GPR((G["g_X"], G["g_Y"][:, :1]), gp.kernels.SquaredExponential(variance=float(G["g_variance"]),
                                                                               lengthscales=G["g_lengthscales"]),
                      noise_variance=float(G["g_noise_variance"]))
    m.kernel.lengthscales.prior = priors.Gamma(2.0, 3.0)
    m.kernel.variance.prior = priors.LogNormal(0.1, 0.8)
    m.likelihood.variance.prior = priors.Normal(-1.0, 2.0)
    m.likelihood.variance.prior_on = PriorOn.UNCONSTRAINED
    np.testing.assert_allclose(m.log_prior_density(), float(G["g_map_log_prior"]), rtol=1e-12)
    np.testing.assert_allclose(float(_np(m.log_posterior_density())), float(G["g_map_log_posterior"]), rtol=1e-10)
    np.testing.assert_allclose(float(_np(m.training_loss())), float(G["g_map_training_loss"]), rtol=1e-10)
    for name, par in (("lengthscales", m.kernel.lengthscales), ("variance", m.kernel.variance), ("noise_variance", m.likelihood.variance)):
        np.testing.assert_allclose(np.asarray(par.unconstrained_variable), G[f"g_map_u_{name}"], rtol=1e-12)
    v, g = m.objective_and_grad()
    np.testing.assert_allclose(v, float(G["g_map_log_posterior"]), rtol=1e-10)
    for name, par in (("lengthscales", m.kernel.lengthscales), ("variance", m.kernel.variance), ("noise_variance", m.likelihood.variance)):
        ref = G[f"g_map_d{name}"]
        assert np.abs(np.asarray(g[par]).reshape(ref.shape) - ref).max() <= 1e-7 * np.abs(ref).max(), name
