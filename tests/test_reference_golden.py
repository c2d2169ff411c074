"""CPU: the oracle (oracle/gp_oracle.py) against values computed by the REFERENCE'S OWN SOURCE.

tests/golden/ref_golden.npz was written by tests/golden/make_golden_ref.py, which imports the unmodified GPflow package
from /root/reference over NumPy stand-ins for TensorFlow / TFP / check_shapes / multipledispatch
(tests/golden/refshim/) and calls its public API.  This file is what pins the oracle to the reference: every function
of the oracle that the parity tests lean on reproduces the reference's numbers here (1e-12 scaled), and -- when the
reference tree is present (the build container, not the GPU box) -- the fixture itself is regenerated and compared.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gp_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_golden.npz"))
TOL = 1e-12


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    assert float(np.max(np.abs(a - b))) <= tol * scale, float(np.max(np.abs(a - b))) / scale


@pytest.mark.skipif(not os.path.isdir("/root/reference/gpflow"), reason="reference tree not on this machine")
def test_fixture_regenerates_from_the_reference_source():
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_golden_ref.py"), "--check"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "reproduces from the reference source" in r.stdout


def test_kernels():
    X, X2, ls, var = G["k_X"], G["k_X2"], G["k_ls"], float(G["k_var"])
    for name in ("SquaredExponential", "Matern12", "Matern32", "Matern52"):
        close(orc.stationary_K(name, X, None, variance=var, lengthscales=ls), G[f"k_{name}_sym"])
        close(orc.stationary_K(name, X, X2, variance=var, lengthscales=ls), G[f"k_{name}_cross"])
        close(orc.stationary_K_diag(X, variance=var), G[f"k_{name}_diag"])
    ad = lambda A: A[:, [1, 3]]  # noqa: E731  active_dims = [1, 3] (kernels/base.py:90-109)
    sl = lambda A: A[:, 0:2]  # noqa: E731
    k_ad = lambda A, B=None: orc.rbf_K(ad(A), None if B is None else ad(B), variance=0.8, lengthscales=np.array([0.5, 1.5]))  # noqa: E731
    k_sl = lambda A, B=None: orc.rbf_K(sl(A), None if B is None else sl(B), variance=0.8, lengthscales=0.6)  # noqa: E731
    close(k_ad(X), G["k_active_dims_sym"])
    close(k_ad(X, X2) + k_sl(X, X2), G["k_sum_cross"])
    close(k_ad(X) * orc.stationary_K("Matern32", X, None, variance=1.1, lengthscales=0.9), G["k_prod_sym"])
    close(np.full(len(X), 1.6), G["k_sum_diag"])


def test_gpr():
    kw = dict(variance=1.0, lengthscales=2.0, noise_variance=1.0)
    X, Y, Xn = G["gpr_X"], G["gpr_Y"], G["gpr_Xnew"]
    close(orc.gpr_log_marginal_likelihood(X, Y, **kw), G["gpr_lml"])
    mu, var = orc.gpr_predict_f(X, Y, Xn, **kw)
    close(mu, G["gpr_mu"]); close(var, G["gpr_var"])
    close(mu, G["gpr_cached_mu"]); close(var, G["gpr_cached_var"], 1e-11)
    _, vfc = orc.gpr_predict_f(X, Y, Xn, full_cov=True, **kw)
    close(vfc, G["gpr_var_fullcov"])
    ymu, yvar = orc.gaussian_predict_mean_and_var(mu, var, 1.0)
    close(ymu, G["gpr_ymu"]); close(yvar, G["gpr_yvar"])
    close(orc.gaussian_predict_log_density(mu, var, np.cos(Xn), 1.0), G["gpr_logdens"])
    # two columns, constant mean, ARD, Xnew with leading batch dims
    kw2 = dict(variance=1.3, lengthscales=G["gpr2_ls"], noise_variance=0.07, mean=0.3)
    X, Y, Xb = G["gpr2_X"], G["gpr2_Y"], G["gpr2_Xnew"]
    close(orc.gpr_log_marginal_likelihood(X, Y, **kw2), G["gpr2_lml"])
    mu, var = orc.gpr_predict_f(X, Y, Xb.reshape(-1, 3), **kw2)
    close(mu.reshape(2, 5, 2), G["gpr2_mu"]); close(var.reshape(2, 5, 2), G["gpr2_var"])
    for b in range(2):  # full_cov over leading batch dims: [..., P, T, T] (conditionals/util.py:108-131)
        _, vfc = orc.gpr_predict_f(X, Y, Xb[b], full_cov=True, **kw2)
        close(vfc, G["gpr2_var_fullcov"][b])
    kw1 = dict(variance=1.0, lengthscales=1.0, noise_variance=0.1)
    close(orc.gpr_log_marginal_likelihood(G["c1_X"], G["c1_Y"], **kw1), G["c1_lml"])
    mu, var = orc.gpr_predict_f(G["c1_X"], G["c1_Y"], G["c1_Xnew"], **kw1)
    close(mu, G["c1_mu"]); close(var, G["c1_var"])


def test_heteroskedastic_gaussian():
    """Gaussian(scale=Linear(A, b)) / Gaussian(variance=Polynomial) of the reference (likelihoods/scalar_continuous.py:52-148,
    utilities/model_utils.py:46-50) behind GPR (LML, predict_f / predict_y / predict_log_density) and SVGP.elbo (both whitenings,
    q_diag): the oracle with one noise variance per row."""
    X, Y, Xs = G["het_X"], G["het_Y"], G["het_Xnew"]
    lin = orc.linear_function(G["het_A"], G["het_b"])
    nv = orc.gaussian_variance_at(X, scale=lin)
    close(nv[:, None], G["het_variance_at"], 1e-15)
    kw = dict(variance=1.3, lengthscales=np.array([0.3, 0.6]))
    close(orc.gpr_log_marginal_likelihood(X, Y, noise_variance=nv, **kw), G["het_gpr_lml"])
    mu, var = orc.gpr_predict_f(X, Y, Xs, noise_variance=nv, **kw)
    close(mu, G["het_gpr_fmu"]); close(var, G["het_gpr_fvar"])
    close(mu, G["het_gpr_cached_mu"]); close(var, G["het_gpr_cached_var"], 1e-11)
    nvs = orc.gaussian_variance_at(Xs, scale=lin)
    ymu, yvar = orc.gaussian_predict_mean_and_var(mu, var, nvs)
    close(ymu, G["het_gpr_ymu"]); close(yvar, G["het_gpr_yvar"])
    close(orc.gaussian_predict_log_density(mu, var, np.cos(Xs[:, :1]), nvs), G["het_gpr_logdens"])
    for wh, name in ((True, "white"), (False, "unwhite")):
        close(orc.svgp_elbo(X, Y, G["het_Z"], G["het_q_mu"], G["het_q_sqrt"], noise_variance=nv, whiten=wh, num_data=400, **kw),
              G[f"het_svgp_elbo_{name}"])
    # predictions through the likelihood at new inputs: sigma^2 evaluated THERE (likelihoods/base.py, scalar_continuous.py:107-148)
    smu_, svar_ = orc.svgp_predict_f(Xs, G["het_Z"], G["het_q_mu"], G["het_q_sqrt"], whiten=True, **kw)
    symu, syvar = orc.gaussian_predict_mean_and_var(smu_, svar_, nvs)
    close(symu, G["het_svgp_ymu"]); close(syvar, G["het_svgp_yvar"])
    close(orc.gaussian_predict_log_density(smu_, svar_, np.cos(Xs[:, :1]), nvs), G["het_svgp_logdens"])
    # variance as a polynomial that dips below the lower bound: clipped at 1e-3 when evaluated (parameter_or_function.py:52-56)
    w = G["het_poly_w"]
    poly = lambda X_: w[0, 0] + w[0, 1] * X_[:, 1:2] + w[0, 2] * X_[:, 1:2] ** 2   # noqa: E731  (powers (0,0), (0,1), (0,2))
    nvp = orc.gaussian_variance_at(X, variance=poly, lower_bound=1e-3)
    close(nvp[:, None], G["het_poly_variance_at"], 1e-15)
    assert (nvp == 1e-3).sum() == 13
    close(orc.gpr_log_marginal_likelihood(X, Y, noise_variance=nvp, **kw), G["het_poly_gpr_lml"])
    close(orc.svgp_elbo(X, Y, G["het_Z"], G["het_q_mu"], G["het_poly_q_sqrt_diag"], noise_variance=nvp, whiten=True, num_data=400,
                        **kw), G["het_poly_svgp_elbo_qdiag"])
    # SGPR with one sigma_n per data row (sgpr.py:181-384)
    skw = dict(noise_variance=nv, **kw)
    close(orc.sgpr_elbo(X, Y, G["het_Z"], **skw), G["het_sgpr_elbo"])
    smu, svar = orc.sgpr_predict_f(X, Y, G["het_Z"], Xs, **skw)
    close(smu, G["het_sgpr_mu"]); close(svar, G["het_sgpr_var"], 1e-11)
    qmu, qcov = orc.sgpr_compute_qu(X, Y, G["het_Z"], **skw)
    close(qmu, G["het_sgpr_qu_mu"], 1e-10); close(qcov, G["het_sgpr_qu_cov"], 1e-10)
    close(orc.sgpr_upper_bound(X, Y, G["het_Z"], **skw), G["het_sgpr_upper"])   # every row rescaled by its own sigma_n^2 + c
    gymu, gyvar = orc.gaussian_predict_mean_and_var(smu, svar, nvs)
    close(gymu, G["het_sgpr_ymu"]); close(gyvar, G["het_sgpr_yvar"], 1e-11)
    close(orc.gaussian_predict_log_density(smu, svar, np.cos(Xs[:, :1]), nvs), G["het_sgpr_logdens"], 1e-11)
    assert float(G["het_sgpr_elbo"]) < float(G["het_gpr_lml"]) < float(G["het_sgpr_upper"])


def test_gauss_kl():
    mu, sq, K, Kb = G["kl_mu"], G["kl_sqrt"], G["kl_K"], G["kl_Kb"]
    close(orc.gauss_kl(mu, sq), G["kl_white"]); close(orc.gauss_kl(mu, sq, K), G["kl_K_val"])
    close(orc.gauss_kl(mu, sq, Kb), G["kl_Kb_val"])
    close(orc.gauss_kl(mu, G["kl_sqrt_diag"]), G["kl_diag_white"]); close(orc.gauss_kl(mu, G["kl_sqrt_diag"], K), G["kl_diag_K"])
    close(orc.gauss_kl(mu, G["kl_sqrt_upper"], K), G["kl_upper_ignored"])
    close(orc.gauss_kl(mu, sq, K_cholesky=np.linalg.cholesky(K)), G["kl_Kchol"])


def _cond(Xn, Z, f, q, white, fc):
    kw = dict(variance=1.4, lengthscales=np.array([0.8, 1.2]))
    Kmm = orc.Kuu(Z, jitter=orc.DEFAULT_JITTER, **kw)
    Kmn = orc.Kuf(Z, Xn, **kw)
    Knn = orc.rbf_K(Xn, None, **kw) if fc else orc.stationary_K_diag(Xn, variance=1.4)
    return orc.base_conditional(Kmn, Kmm, Knn, f, full_cov=fc, q_sqrt=q, white=white)


def test_conditionals():
    Z, Xn, f = G["cond_Z"], G["cond_X"], G["cond_f"]
    for white in (False, True):
        for fc in (False, True):
            for tag, q in (("full", G["cond_qs"]), ("diag", G["cond_qd"]), ("none", None)):
                mu, var = _cond(Xn, Z, f, q, white, fc)
                close(mu, G[f"cond_w{int(white)}_fc{int(fc)}_{tag}_mu"]); close(var, G[f"cond_w{int(white)}_fc{int(fc)}_{tag}_var"])
    Xb = G["cond_Xb"]
    for white, key in ((True, "cond_batch"), (False, "cond_batch_unw")):
        for fc in (False, True):
            for i in range(2):
                for j in range(3):
                    mu, var = _cond(Xb[i, j], Z, f, G["cond_qs"], white, fc)
                    close(mu, G[f"{key}_fc{int(fc)}_mu"][i, j]); close(var, G[f"{key}_fc{int(fc)}_var"][i, j])
    mu, var = orc.base_conditional(G["bc_Kmn"], G["bc_Kmm"], G["bc_Knn"], f, q_sqrt=G["cond_qs"], white=False)
    close(mu, G["bc_mu"]); close(var, G["bc_var"])


def test_svgp():
    X, Y, Z, q_mu, q_sqrt, Xs = (G[k] for k in ("svgp_X", "svgp_Y", "svgp_Z", "svgp_q_mu", "svgp_q_sqrt", "svgp_Xnew"))
    kw = dict(variance=1.0, lengthscales=1.0)
    for w in (0, 1):
        close(orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, noise_variance=1.0, whiten=bool(w), **kw), G[f"svgp_elbo_w{w}"])
        close(orc.prior_kl(Z, q_mu, q_sqrt, whiten=bool(w), **kw), G[f"svgp_kl_w{w}"])
        mu, var = orc.svgp_predict_f(Xs, Z, q_mu, q_sqrt, whiten=bool(w), **kw)
        close(mu, G[f"svgp_mu_w{w}"]); close(var, G[f"svgp_var_w{w}"])
        _, vfc = orc.svgp_predict_f(Xs, Z, q_mu, q_sqrt, whiten=bool(w), full_cov=True, **kw)
        close(vfc, G[f"svgp_var_fullcov_w{w}"])
        alpha, Qinv = orc.svgp_precompute(Z, q_mu, q_sqrt, whiten=bool(w), **kw)
        pmu, pvar = orc.svgp_predict_with_precompute(alpha, Qinv, Z, Xs, **kw)
        close(pmu, G[f"svgp_cached_mu_w{w}"], 1e-10); close(pvar, G[f"svgp_cached_var_w{w}"], 1e-10)
        close(orc.svgp_elbo(X, Y, Z, q_mu, G["svgp_q_sqrt_diag"], noise_variance=1.0, whiten=bool(w), num_data=100, **kw),
              G[f"svgp_elbo_diag_w{w}"])
        close(orc.svgp_elbo(G["mid_X"], G["mid_Y"], G["mid_Z"], G["mid_q_mu"], G["mid_q_sqrt"], variance=1.0, lengthscales=G["mid_ls"],
                            noise_variance=0.1, whiten=bool(w), num_data=100000, mean=0.2), G[f"mid_elbo_w{w}"])


def test_multi_output():
    X, Y, Z, Xs, q_mu, q_sqrt = (G[k] for k in ("mo_X", "mo_Y", "mo_Z", "mo_Xnew", "mo_q_mu", "mo_q_sqrt"))
    kw = dict(variance=1.2, lengthscales=np.array([0.9, 1.1]))
    L = q_mu.shape[1]
    for w in (0, 1):
        close(orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, noise_variance=0.2, whiten=bool(w), **kw), G[f"mo_shared_elbo_w{w}"])
        mu, var = orc.svgp_predict_f(Xs, Z, q_mu, q_sqrt, whiten=bool(w), **kw)
        close(mu, G[f"mo_shared_mu_w{w}"]); close(var, G[f"mo_shared_var_w{w}"])
        alpha, Qinv = orc.svgp_precompute(Z, q_mu, q_sqrt, whiten=bool(w), **kw)
        pmu, pvar = orc.svgp_predict_with_precompute(alpha, Qinv, Z, Xs, **kw)
        close(pmu, G[f"mo_shared_cached_mu_w{w}"], 1e-10); close(pvar, G[f"mo_shared_cached_var_w{w}"], 1e-10)
    lss, vars_, Zs = G["mo_sep_ls"], G["mo_sep_var"], G["mo_sep_Z"]
    for w in (0, 1):
        close(orc.svgp_elbo_separate(X, Y, Zs, q_mu, q_sqrt, variances=vars_, lengthscales_list=lss, noise_variance=0.2,
                                     whiten=bool(w)), G[f"mo_sep_elbo_w{w}"])
        for fc in (0, 1):
            mus, vs, pmus, pvs = [], [], [], []
            for p in range(L):  # one single-output GP per latent (conditionals/util.py:566-629; posteriors.py:862-887)
                kwp = dict(variance=vars_[p], lengthscales=lss[p])
                mu, var = orc.svgp_predict_f(Xs, Zs[p], q_mu[:, p:p + 1], q_sqrt[p:p + 1], whiten=bool(w), full_cov=bool(fc), **kwp)
                mus.append(mu[:, 0]); vs.append(var[0] if fc else var[:, 0])
                alpha, Qinv = orc.svgp_precompute(Zs[p], q_mu[:, p:p + 1], q_sqrt[p:p + 1], whiten=bool(w), **kwp)
                pmu, pv = orc.svgp_predict_with_precompute(alpha, Qinv, Zs[p], Xs, full_cov=bool(fc), **kwp)
                pmus.append(pmu[:, 0]); pvs.append(pv[0] if fc else pv[:, 0])
            stack_v = (lambda v: np.stack(v, 0)) if fc else (lambda v: np.stack(v, 1))
            close(np.stack(mus, 1), G[f"mo_sep_mu_w{w}_fc{fc}"]); close(stack_v(vs), G[f"mo_sep_var_w{w}_fc{fc}"])
            close(np.stack(pmus, 1), G[f"mo_sep_cached_mu_w{w}_fc{fc}"], 1e-10); close(stack_v(pvs), G[f"mo_sep_cached_var_w{w}_fc{fc}"], 1e-10)
        # full_output_cov of independent outputs: diagonal embedding [T, P, P] (conditionals/util.py:222-239)
        foc = G[f"mo_sep_var_w{w}_foc"]
        diag = G[f"mo_sep_var_w{w}_fc0"]
        close(np.stack([np.diag(diag[t]) for t in range(diag.shape[0])]), foc)


def test_sgpr():
    X, Y, Z, Xs = G["sgpr_X"], G["sgpr_Y"], G["sgpr_Z"], G["sgpr_Xnew"]
    kw = dict(variance=1.1, lengthscales=np.array([0.8, 1.2]), noise_variance=0.05)
    close(orc.sgpr_elbo(X, Y, Z, **kw), G["sgpr_elbo"]); close(orc.sgpr_upper_bound(X, Y, Z, **kw), G["sgpr_upper"])
    mu, var = orc.sgpr_predict_f(X, Y, Z, Xs, **kw)
    close(mu, G["sgpr_mu"]); close(var, G["sgpr_var"], 1e-11)
    qmu, qcov = orc.sgpr_compute_qu(X, Y, Z, **kw)
    close(qmu, G["sgpr_qu_mu"], 1e-10); close(qcov, G["sgpr_qu_cov"], 1e-10)


def test_gradient_oracle_pinned_to_reference_finite_differences():
    """oracle/gp_oracle_grad.py (torch autograd over the restated algorithm) against gradients of the REFERENCE'S OWN
    SVGP.elbo / GPR.log_marginal_likelihood / SGPR.elbo, taken by Richardson-extrapolated central differences of the
    unmodified GPflow source (tests/golden/make_golden_ref.py; what optimizers/scipy.py:174-221, 322-331 differentiates
    with tf.GradientTape).  1e-7 of each gradient's largest entry: the finite differences themselves carry ~1e-8."""
    from oracle import gp_oracle_grad as og
    X, Y, Z, qm, qs = G["g_X"], G["g_Y"], G["g_Z"], G["g_q_mu"], G["g_q_sqrt"]
    kw = dict(variance=float(G["g_variance"]), lengthscales=G["g_lengthscales"], noise_variance=float(G["g_noise_variance"]))

    def chk(g, ref, tol=1e-7):
        g = np.asarray(g, dtype=np.float64).reshape(ref.shape)
        assert np.abs(g - ref).max() <= tol * max(np.abs(ref).max(), 1e-300), (np.abs(g - ref).max(), np.abs(ref).max())
    for w in (1, 0):
        F, g = og.svgp_elbo_value_and_grads(X, Y, Z, qm, qs, num_data=500, whiten=bool(w), **kw)
        np.testing.assert_allclose(F, float(G[f"g_svgp_elbo_w{w}"]), rtol=1e-12)
        for n in ("variance", "lengthscales", "noise_variance", "Z", "q_mu"):
            chk(g[n], G[f"g_svgp_d{n}_w{w}"])
        chk(np.tril(g["q_sqrt"]), G[f"g_svgp_dq_sqrt_w{w}"])
    F, g = og.gpr_lml_value_and_grads(X, Y[:, :1], **kw)
    np.testing.assert_allclose(F, float(G["g_gpr_lml"]), rtol=1e-12)
    for n in ("variance", "lengthscales", "noise_variance"):
        chk(g[n], G[f"g_gpr_d{n}"])
    F, g = og.sgpr_elbo_value_and_grads(X, Y[:, :1], Z, **kw)
    np.testing.assert_allclose(F, float(G["g_sgpr_elbo"]), rtol=1e-12)
    for n in ("variance", "lengthscales", "noise_variance", "Z"):
        chk(g[n], G[f"g_sgpr_d{n}"])


def _nested_members():
    members = [("SquaredExponential", float(G["nest_v0"]), G["nest_ls0"]), ("Matern32", float(G["nest_v1"]), np.array(0.8)),
               ("Matern52", float(G["nest_v2"]), np.array(float(G["nest_ls2"])))]
    return members, ("mul", [("add", [0, 1]), 2]), [None, [1], None]


def test_nested_combination_oracle_pinned_to_reference():
    """(SquaredExponential + Matern32[dim 1]) * Matern52 (kernels/base.py:223-329: a Product holding a Sum): the autograd oracle's
    GPR LML / whitened SVGP ELBO and their gradients w.r.t. the members' variances and lengthscales against the reference's own
    forward code and its Richardson differences."""
    from oracle import gp_oracle_grad as og
    members, tree, cols = _nested_members()
    X, Y, Z, qm, qs = G["g_X"], G["g_Y"], G["g_Z"], G["g_q_mu"], G["g_q_sqrt"]
    v, g = og.combination_value_and_grads("gpr", X, Y[:, :1], members, tree, noise_variance=0.15, cols=cols)
    np.testing.assert_allclose(v, float(G["nest_gpr_lml"]), rtol=1e-12)
    v2, g2 = og.combination_value_and_grads("svgp", X, Y, members, tree, noise_variance=0.15, Z=Z, q_mu=qm, q_sqrt=qs, num_data=500, cols=cols)
    np.testing.assert_allclose(v2, float(G["nest_svgp_elbo"]), rtol=1e-12)
    for gg, pre in ((g, "nest_gpr"), (g2, "nest_svgp")):
        for i in range(3):
            ref = float(G[f"{pre}_dv{i}"])
            assert abs(gg["variance"][i] - ref) <= 1e-7 * max(1.0, abs(ref)), (pre, i)
        for i, nm in ((0, "ls0"), (2, "ls2")):
            ref = np.atleast_1d(G[f"{pre}_d{nm}"])
            assert np.abs(np.ravel(gg["lengthscales"][i]) - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), (pre, nm)


def test_natgrad_conversions_pinned_to_reference():
    """The natural-gradient parameter conversions restated in oracle/gp_oracle_grad.py (which the natural-gradient oracle
    differentiates) against the reference's own functions run through the shim (optimizers/natgrad.py:429-516)."""
    from oracle import gp_oracle_grad as og
    c = og.natgrad_conversions(G["ng_mu"], G["ng_sqrt"])
    for k in ("nat1", "nat2", "eta1", "eta2", "back_mu", "back_sqrt", "back2_mu", "back2_sqrt"):
        ref = G[f"ng_{k}"]
        np.testing.assert_allclose(np.asarray(c[k]).reshape(ref.shape), ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()), err_msg=k)
    np.testing.assert_allclose(c["back_mu"], G["ng_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(c["back_sqrt"], G["ng_sqrt"], rtol=0, atol=1e-10)

