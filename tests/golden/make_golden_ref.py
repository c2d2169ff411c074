"""Golden vectors computed BY THE REFERENCE'S OWN SOURCE (the unmodified GPflow package under /root/reference), executed
in the build container over the NumPy stand-ins of tests/golden/refshim/ for TensorFlow, TensorFlow-Probability,
check_shapes, multipledispatch and deprecated (none of which is installable there; see refshim/README.md).

    python tests/golden/make_golden_ref.py            # writes tests/golden/ref_golden.npz
    python tests/golden/make_golden_ref.py --check    # recompute and compare with the committed file (1e-13)

Every value below comes out of the reference's public API -- gpflow.kernels.*, gpflow.models.GPR / SVGP / SGPR,
gpflow.conditionals.conditional / base_conditional, gpflow.kullback_leiblers.gauss_kl, gpflow.posteriors (fused and
cached), gpflow.covariances.Kuu / Kuf, multi-output kernels and inducing variables -- i.e. the statements of
gpflow/conditionals/util.py, kullback_leiblers.py, posteriors.py, models/*.py are the ones that run; only the array
primitives underneath are NumPy / LAPACK instead of TensorFlow / Eigen.  The oracle (oracle/gp_oracle.py) and the HIP
path are then tested against this file (tests/test_reference_golden.py, tests/test_gpu_reference_golden.py).

Fixtures: the reference's own test fixtures where they are seedable (tests/gpflow/models/test_gpr.py:21-30,
tests/gpflow/models/test_svgp.py:28-36, tests/gpflow/test_kullback_leiblers.py:106-118,
tests/gpflow/posteriors/test_posteriors.py:346-388, tests/gpflow/conditionals/test_broadcasted_conditionals.py:58-150)
and seeded random cases at the sizes of BASELINE config C1.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GPFLOW_REFERENCE", "/root/reference")


def _import_reference():
    if not os.path.isdir(os.path.join(REF, "gpflow")):
        raise SystemExit(f"reference tree not found at {REF}")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(HERE, "refshim"))
    import gpflow  # noqa: F401  (the REAL reference source)
    import tensorflow as tf  # noqa: F401  (the stand-in)
    assert os.path.abspath(gpflow.__file__).startswith(os.path.abspath(REF)), gpflow.__file__
    return gpflow


def _n(x):
    return np.asarray(x.numpy() if hasattr(x, "numpy") else x, dtype=np.float64)


def build():  # noqa: C901
    gpflow = _import_reference()
    import tensorflow as tf
    from gpflow.conditionals import base_conditional, conditional
    from gpflow.kullback_leiblers import gauss_kl
    out = {}

    def T(a):  # (the reference's own tests hand tf tensors to the functional API)
        return None if a is None else tf.convert_to_tensor(np.asarray(a, dtype=np.float64))

    # ---- kernels (kernels/stationaries.py, kernels/base.py) ---------------------------------------------------------
    rng = np.random.default_rng(11)
    X = rng.normal(size=(23, 4)); X2 = rng.normal(size=(9, 4))
    ls = np.array([0.7, 1.3, 0.9, 2.1])
    out.update(k_X=X, k_X2=X2, k_ls=ls, k_var=1.7)
    for name in ("SquaredExponential", "Matern12", "Matern32", "Matern52"):
        k = getattr(gpflow.kernels, name)(variance=1.7, lengthscales=ls)
        out[f"k_{name}_sym"] = _n(k(X)); out[f"k_{name}_cross"] = _n(k(X, X2)); out[f"k_{name}_diag"] = _n(k(X, full_cov=False))
    k_ad = gpflow.kernels.SquaredExponential(variance=0.8, lengthscales=[0.5, 1.5], active_dims=[1, 3])
    out["k_active_dims_sym"] = _n(k_ad(X))
    k_sl = gpflow.kernels.SquaredExponential(variance=0.8, lengthscales=0.6, active_dims=slice(0, 2))
    ksum = k_ad + k_sl
    kprod = k_ad * gpflow.kernels.Matern32(variance=1.1, lengthscales=0.9)
    out["k_sum_cross"] = _n(ksum(X, X2)); out["k_prod_sym"] = _n(kprod(X)); out["k_sum_diag"] = _n(ksum(X, full_cov=False))

    # ---- GPR (models/gpr.py, logdensities.py, posteriors.py:361-443) ------------------------------------------------
    rng = np.random.RandomState(0)
    X = rng.randn(10, 1); Y = np.sin(X) + 0.1 * rng.randn(10, 1); Xnew = rng.randn(7, 1)
    m = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=2.0), noise_variance=1.0)
    mu, var = m.predict_f(Xnew); mu_fc, var_fc = m.predict_f(Xnew, full_cov=True)
    ymu, yvar = m.predict_y(Xnew)
    out.update(gpr_X=X, gpr_Y=Y, gpr_Xnew=Xnew, gpr_lml=float(m.log_marginal_likelihood()), gpr_mu=_n(mu), gpr_var=_n(var),
               gpr_var_fullcov=_n(var_fc), gpr_ymu=_n(ymu), gpr_yvar=_n(yvar),
               gpr_logdens=_n(m.predict_log_density((Xnew, np.cos(Xnew)))))
    post = m.posterior()
    pmu, pvar = post.predict_f(Xnew)
    out.update(gpr_cached_mu=_n(pmu), gpr_cached_var=_n(pvar))
    # two output columns + constant mean + ARD, D = 3 (gpr.py:105-107: LML is the sum over columns)
    rng = np.random.default_rng(5)
    X = rng.normal(size=(40, 3)); Y = np.stack([np.sin(X.sum(1)), np.cos(X[:, 0])], 1) + 0.05 * rng.normal(size=(40, 2))
    Xnew = rng.normal(size=(2, 5, 3))  # leading batch dims (tests/gpflow/models/test_gpr_posterior.py:36-39)
    ls3 = np.array([0.9, 1.4, 0.6])
    m2 = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(variance=1.3, lengthscales=ls3),
                           mean_function=gpflow.mean_functions.Constant(np.array([0.3])), noise_variance=0.07)
    mu, var = m2.predict_f(Xnew)
    mu_fc, var_fc = m2.predict_f(Xnew, full_cov=True)
    out.update(gpr2_X=X, gpr2_Y=Y, gpr2_Xnew=Xnew, gpr2_ls=ls3, gpr2_lml=float(m2.log_marginal_likelihood()), gpr2_mu=_n(mu),
               gpr2_var=_n(var), gpr2_var_fullcov=_n(var_fc))
    # BASELINE config C1: N = 512, D = 2
    rng = np.random.default_rng(1)
    X = rng.normal(size=(512, 2)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(512, 1))
    m1 = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(), noise_variance=0.1)
    Xt = rng.normal(size=(33, 2))
    mu, var = m1.predict_f(Xt)
    out.update(c1_X=X, c1_Y=Y, c1_Xnew=Xt, c1_lml=float(m1.log_marginal_likelihood()), c1_mu=_n(mu), c1_var=_n(var))

    # ---- gauss_kl (kullback_leiblers.py:59-165) -----------------------------------------------------------------------
    rng = np.random.RandomState(0)
    M = 5
    mu = rng.randn(M, 4); sqrt = np.array([np.tril(rng.randn(M, M)) for _ in range(4)])
    A = rng.randn(M, M); K = A @ A.T + 1e-6 * np.eye(M)
    Kb = np.stack([K + 0.1 * i * np.eye(M) for i in range(4)])
    sq_diag = np.abs(rng.randn(M, 4)) + 0.2
    full_with_upper = sqrt + np.triu(rng.randn(4, M, M), 1)  # upper part must be ignored (band_part, :120)
    out.update(kl_mu=mu, kl_sqrt=sqrt, kl_K=K, kl_Kb=Kb, kl_sqrt_diag=sq_diag, kl_sqrt_upper=full_with_upper,
               kl_white=float(gauss_kl(T(mu), T(sqrt))), kl_K_val=float(gauss_kl(T(mu), T(sqrt), T(K))), kl_Kb_val=float(gauss_kl(T(mu), T(sqrt), T(Kb))),
               kl_diag_white=float(gauss_kl(T(mu), T(sq_diag))), kl_diag_K=float(gauss_kl(T(mu), T(sq_diag), T(K))),
               kl_upper_ignored=float(gauss_kl(T(mu), T(full_with_upper), T(K))),
               kl_Kchol=float(gauss_kl(T(mu), T(sqrt), K_cholesky=T(np.linalg.cholesky(K)))))

    # ---- base_conditional / conditional (conditionals/util.py:37-169, conditionals.py) -----------------------------------
    rng = np.random.default_rng(21)
    M, N, R = 6, 11, 3
    Z = rng.normal(size=(M, 2)); Xn = rng.normal(size=(N, 2))
    kern = gpflow.kernels.SquaredExponential(variance=1.4, lengthscales=[0.8, 1.2])
    f = rng.normal(size=(M, R)); qs = np.tril(rng.normal(size=(R, M, M))) * 0.3 + 0.5 * np.eye(M)
    qd = np.abs(rng.normal(size=(M, R))) + 0.1
    out.update(cond_Z=Z, cond_X=Xn, cond_f=f, cond_qs=qs, cond_qd=qd)
    iv = gpflow.inducing_variables.InducingPoints(Z)
    for white in (False, True):
        for fc in (False, True):
            for tag, q in (("full", qs), ("diag", qd), ("none", None)):
                mu, var = conditional(T(Xn), iv, kern, T(f), full_cov=fc, q_sqrt=T(q), white=white)
                out[f"cond_w{int(white)}_fc{int(fc)}_{tag}_mu"] = _n(mu); out[f"cond_w{int(white)}_fc{int(fc)}_{tag}_var"] = _n(var)
    # leading batch dims on Xnew, with and without full_cov (test_broadcasted_conditionals.py:58-150; util.py:108-131)
    Xb = rng.normal(size=(2, 3, N, 2))
    out["cond_Xb"] = Xb
    for fc in (False, True):
        mu, var = conditional(T(Xb), iv, kern, T(f), full_cov=fc, q_sqrt=T(qs), white=True)
        out[f"cond_batch_fc{int(fc)}_mu"] = _n(mu); out[f"cond_batch_fc{int(fc)}_var"] = _n(var)
        mu, var = conditional(T(Xb), iv, kern, T(f), full_cov=fc, q_sqrt=T(qs), white=False)
        out[f"cond_batch_unw_fc{int(fc)}_mu"] = _n(mu); out[f"cond_batch_unw_fc{int(fc)}_var"] = _n(var)
    Kmm = _n(kern(Z)) + 1e-6 * np.eye(M); Kmn = _n(kern(Z, Xn)); Knn = _n(kern(Xn, full_cov=False))
    mu, var = base_conditional(T(Kmn), T(Kmm), T(Knn), T(f), full_cov=False, q_sqrt=T(qs), white=False)
    out.update(bc_Kmm=Kmm, bc_Kmn=Kmn, bc_Knn=Knn, bc_mu=_n(mu), bc_var=_n(var))

    # ---- SVGP (models/svgp.py; reference fixture tests/gpflow/models/test_svgp.py:28-36) -----------------------------------
    rng = np.random.RandomState(0)
    X = rng.randn(20, 1); Y = rng.randn(20, 2) ** 2; Z = rng.randn(3, 1)
    q_mu = rng.randn(3, 2); q_sqrt = np.array([np.tril(rng.randn(3, 3)) for _ in range(2)])
    q_sqrt[:, np.arange(3), np.arange(3)] = np.abs(q_sqrt[:, np.arange(3), np.arange(3)]) + 0.1
    q_sd = np.abs(rng.randn(3, 2)) + 0.3
    Xs = rng.randn(6, 1)
    out.update(svgp_X=X, svgp_Y=Y, svgp_Z=Z, svgp_q_mu=q_mu, svgp_q_sqrt=q_sqrt, svgp_q_sqrt_diag=q_sd, svgp_Xnew=Xs)
    for w in (0, 1):
        s = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=1.0), gpflow.likelihoods.Gaussian(variance=1.0),
                               Z.copy(), q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), whiten=bool(w), num_latent_gps=2)
        out[f"svgp_elbo_w{w}"] = float(s.elbo((X, Y))); out[f"svgp_kl_w{w}"] = float(s.prior_kl())
        mu, var = s.predict_f(Xs); mu2, var_fc = s.predict_f(Xs, full_cov=True)
        out[f"svgp_mu_w{w}"] = _n(mu); out[f"svgp_var_w{w}"] = _n(var); out[f"svgp_var_fullcov_w{w}"] = _n(var_fc)
        pmu, pvar = s.posterior().predict_f(Xs)  # cached route (posteriors.py:694-746, 794-822)
        out[f"svgp_cached_mu_w{w}"] = _n(pmu); out[f"svgp_cached_var_w{w}"] = _n(pvar)
        sd = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=1.0), gpflow.likelihoods.Gaussian(variance=1.0),
                                Z.copy(), q_mu=q_mu.copy(), q_sqrt=q_sd.copy(), q_diag=True, whiten=bool(w), num_latent_gps=2, num_data=100)
        out[f"svgp_elbo_diag_w{w}"] = float(sd.elbo((X, Y)))
    # a mid-size case: M = 40, B = 160, D = 3, ARD, one latent, num_data scaling, constant mean
    rng = np.random.default_rng(31)
    X = rng.normal(size=(160, 3)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(160, 1))
    Z = X[:40] + 0.01 * rng.normal(size=(40, 3))
    q_mu = 0.1 * rng.normal(size=(40, 1)); q_sqrt = (np.tril(0.05 * rng.normal(size=(1, 40, 40))) + 0.5 * np.eye(40))
    lsm = np.sqrt(3) * np.array([0.8, 0.85, 0.9])
    out.update(mid_X=X, mid_Y=Y, mid_Z=Z, mid_q_mu=q_mu, mid_q_sqrt=q_sqrt, mid_ls=lsm)
    for w in (0, 1):
        s = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=lsm), gpflow.likelihoods.Gaussian(variance=0.1),
                               Z.copy(), q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(), whiten=bool(w), num_data=100000,
                               mean_function=gpflow.mean_functions.Constant(np.array([0.2])))
        out[f"mid_elbo_w{w}"] = float(s.elbo((X, Y)))

    # ---- multi-output (BASELINE config C5): SharedIndependent / SeparateIndependent ----------------------------------------
    rng = np.random.default_rng(41)
    M, B, D, L = 12, 50, 2, 4
    X = rng.normal(size=(B, D)); Y = rng.normal(size=(B, L)); Z = rng.normal(size=(M, D)); Xs = rng.normal(size=(7, D))
    q_mu = 0.3 * rng.normal(size=(M, L)); q_sqrt = np.tril(0.2 * rng.normal(size=(L, M, M))) + 0.6 * np.eye(M)
    out.update(mo_X=X, mo_Y=Y, mo_Z=Z, mo_Xnew=Xs, mo_q_mu=q_mu, mo_q_sqrt=q_sqrt)
    mo = gpflow.kernels.SharedIndependent(gpflow.kernels.SquaredExponential(variance=1.2, lengthscales=[0.9, 1.1]), output_dim=L)
    ivs = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Z.copy()))
    for w in (0, 1):
        s = gpflow.models.SVGP(mo, gpflow.likelihoods.Gaussian(variance=0.2), ivs, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(),
                               whiten=bool(w), num_latent_gps=L)
        out[f"mo_shared_elbo_w{w}"] = float(s.elbo((X, Y)))
        mu, var = s.predict_f(Xs); out[f"mo_shared_mu_w{w}"] = _n(mu); out[f"mo_shared_var_w{w}"] = _n(var)
        pmu, pvar = s.posterior().predict_f(Xs); out[f"mo_shared_cached_mu_w{w}"] = _n(pmu); out[f"mo_shared_cached_var_w{w}"] = _n(pvar)
    sep_ls = [np.array([0.6 + 0.2 * i, 1.0 + 0.1 * i]) for i in range(L)]
    sep_var = [1.0 + 0.25 * i for i in range(L)]
    Zs = [Z + 0.05 * i for i in range(L)]
    out.update(mo_sep_ls=np.stack(sep_ls), mo_sep_var=np.array(sep_var), mo_sep_Z=np.stack(Zs))
    ksep = gpflow.kernels.SeparateIndependent([gpflow.kernels.SquaredExponential(variance=sep_var[i], lengthscales=sep_ls[i]) for i in range(L)])
    ivsep = gpflow.inducing_variables.SeparateIndependentInducingVariables([gpflow.inducing_variables.InducingPoints(z.copy()) for z in Zs])
    for w in (0, 1):
        s = gpflow.models.SVGP(ksep, gpflow.likelihoods.Gaussian(variance=0.2), ivsep, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(),
                               whiten=bool(w), num_latent_gps=L)
        out[f"mo_sep_elbo_w{w}"] = float(s.elbo((X, Y)))
        for fc in (0, 1):
            mu, var = s.predict_f(Xs, full_cov=bool(fc))
            out[f"mo_sep_mu_w{w}_fc{fc}"] = _n(mu); out[f"mo_sep_var_w{w}_fc{fc}"] = _n(var)
            # the cached separate-kernel posterior (posteriors.py:694-746, 794-822; test_posteriors.py:346-388)
            pmu, pvar = s.posterior().predict_f(Xs, full_cov=bool(fc))
            out[f"mo_sep_cached_mu_w{w}_fc{fc}"] = _n(pmu); out[f"mo_sep_cached_var_w{w}_fc{fc}"] = _n(pvar)
        mu, var = s.predict_f(Xs, full_output_cov=True)
        out[f"mo_sep_var_w{w}_foc"] = _n(var)
    # shared KERNEL over separate inducing variables (posteriors.py:794-822: Kff [N] / [N,N] broadcast over the latents,
    # Kuf [L,M,N]); fused and cached, marginals and full covariance
    for w in (0, 1):
        s = gpflow.models.SVGP(mo, gpflow.likelihoods.Gaussian(variance=0.2), ivsep, q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(),
                               whiten=bool(w), num_latent_gps=L)
        out[f"mo_shsep_elbo_w{w}"] = float(s.elbo((X, Y)))
        for fc in (0, 1):
            mu, var = s.predict_f(Xs, full_cov=bool(fc))
            out[f"mo_shsep_mu_w{w}_fc{fc}"] = _n(mu); out[f"mo_shsep_var_w{w}_fc{fc}"] = _n(var)
            pmu, pvar = s.posterior().predict_f(Xs, full_cov=bool(fc))
            out[f"mo_shsep_cached_mu_w{w}_fc{fc}"] = _n(pmu); out[f"mo_shsep_cached_var_w{w}_fc{fc}"] = _n(pvar)

    # ---- gradients of the reference's own objectives (what optimizers/scipy.py:174-221, 322-331 differentiates) ------------
    # tf.GradientTape is not emulated by the stand-ins, so the gradients are taken by Richardson-extrapolated central
    # differences of the REFERENCE'S forward code (SVGP.elbo, GPR.log_marginal_likelihood, SGPR.elbo) with respect to the
    # constrained quantities -- kernel variance, ARD lengthscales, noise variance, Z, q_mu, tril(q_sqrt) -- each model
    # rebuilt from perturbed values.  (h = 2e-3 and h / 2: truncation O(h^4), rounding ~1e-11; the oracle's autograd and the
    # HIP reverse pass are then tested against these numbers at 1e-7 of the largest entry.)
    def richardson(f, x, h=2e-3):
        x = np.array(x, dtype=np.float64)
        g = np.zeros_like(x)
        it = np.nditer(x, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            d = []
            for hh in (h, h / 2):
                xp, xm = x.copy(), x.copy()
                xp[i] += hh; xm[i] -= hh
                d.append((f(xp) - f(xm)) / (2 * hh))
            g[i] = (4.0 * d[1] - d[0]) / 3.0
        return g

    rng = np.random.default_rng(61)
    Mg, Bg, Dg, Pg = 7, 19, 2, 2
    Xg = rng.normal(size=(Bg, Dg)); Yg = np.sin(Xg.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(Bg, Pg))
    Zg = Xg[:Mg] + 0.05 * rng.normal(size=(Mg, Dg))
    qmg = 0.3 * rng.normal(size=(Mg, Pg)); qsg = np.tril(0.2 * rng.normal(size=(Pg, Mg, Mg))) + 0.6 * np.eye(Mg)
    th = dict(variance=np.array(1.3), lengthscales=np.array([0.9, 1.4]), noise_variance=np.array(0.15))
    out.update(g_X=Xg, g_Y=Yg, g_Z=Zg, g_q_mu=qmg, g_q_sqrt=qsg, g_variance=th["variance"], g_lengthscales=th["lengthscales"],
               g_noise_variance=th["noise_variance"])
    tril_mask = np.tril(np.ones((Mg, Mg), dtype=bool))

    def svgp_elbo_of(whiten, **kw):
        v = dict(Z=Zg, q_mu=qmg, q_sqrt=qsg, **th); v.update(kw)
        mdl = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=float(v["variance"]), lengthscales=np.array(v["lengthscales"])),
                                 gpflow.likelihoods.Gaussian(variance=float(v["noise_variance"])), np.array(v["Z"]),
                                 q_mu=np.array(v["q_mu"]), q_sqrt=np.array(v["q_sqrt"]), whiten=bool(whiten), num_latent_gps=Pg, num_data=500)
        return float(mdl.elbo((Xg, Yg)))

    for w in (1, 0):
        out[f"g_svgp_elbo_w{w}"] = svgp_elbo_of(w)
        for name in ("variance", "lengthscales", "noise_variance"):
            out[f"g_svgp_d{name}_w{w}"] = richardson(lambda x, n=name: svgp_elbo_of(w, **{n: x}), th[name])
        out[f"g_svgp_dZ_w{w}"] = richardson(lambda x: svgp_elbo_of(w, Z=x), Zg)
        out[f"g_svgp_dq_mu_w{w}"] = richardson(lambda x: svgp_elbo_of(w, q_mu=x), qmg)
        gq = richardson(lambda x: svgp_elbo_of(w, q_sqrt=x), qsg)
        out[f"g_svgp_dq_sqrt_w{w}"] = gq * tril_mask[None]   # (the strict upper triangle is ignored by band_part: exact zeros)

    def gpr_lml_of(**kw):
        v = dict(th); v.update(kw)
        mdl = gpflow.models.GPR((Xg, Yg[:, :1]), gpflow.kernels.SquaredExponential(variance=float(v["variance"]),
                                                                                 lengthscales=np.array(v["lengthscales"])),
                                noise_variance=float(v["noise_variance"]))
        return float(mdl.log_marginal_likelihood())
    out["g_gpr_lml"] = gpr_lml_of()
    for name in ("variance", "lengthscales", "noise_variance"):
        out[f"g_gpr_d{name}"] = richardson(lambda x, n=name: gpr_lml_of(**{n: x}), th[name])

    def sgpr_elbo_of(**kw):
        v = dict(Z=Zg, **th); v.update(kw)
        mdl = gpflow.models.SGPR((Xg, Yg[:, :1]), gpflow.kernels.SquaredExponential(variance=float(v["variance"]),
                                                                                  lengthscales=np.array(v["lengthscales"])),
                                 np.array(v["Z"]), noise_variance=float(v["noise_variance"]))
        return float(mdl.elbo())
    out["g_sgpr_elbo"] = sgpr_elbo_of()
    for name in ("variance", "lengthscales", "noise_variance"):
        out[f"g_sgpr_d{name}"] = richardson(lambda x, n=name: sgpr_elbo_of(**{n: x}), th[name])
    out["g_sgpr_dZ"] = richardson(lambda x: sgpr_elbo_of(Z=x), Zg)

    # ---- a NESTED kernel combination, (SquaredExponential + Matern32[dim 1]) * Matern52 (kernels/base.py:223-329): GPR LML and the
    # whitened SVGP ELBO with Richardson gradients w.r.t. the three variances and the lengthscales of the first and last member
    nth = dict(v0=np.array(1.3), v1=np.array(0.6), v2=np.array(0.9), ls0=np.array([0.9, 1.4]), ls2=np.array(1.2))
    out.update({f"nest_{k}": v for k, v in nth.items()})

    def nest_kernel(v):
        return (gpflow.kernels.SquaredExponential(variance=float(v["v0"]), lengthscales=np.array(v["ls0"]))
                + gpflow.kernels.Matern32(variance=float(v["v1"]), lengthscales=0.8, active_dims=[1])) \
            * gpflow.kernels.Matern52(variance=float(v["v2"]), lengthscales=float(v["ls2"]))

    def nest_gpr(**kw):
        v = dict(nth); v.update(kw)
        return float(gpflow.models.GPR((Xg, Yg[:, :1]), nest_kernel(v), noise_variance=0.15).log_marginal_likelihood())

    def nest_svgp(**kw):
        v = dict(nth); v.update(kw)
        mdl = gpflow.models.SVGP(nest_kernel(v), gpflow.likelihoods.Gaussian(variance=0.15), Zg.copy(), q_mu=qmg, q_sqrt=qsg, num_latent_gps=Pg,
                                 num_data=500)
        return float(mdl.elbo((Xg, Yg)))
    out["nest_gpr_lml"], out["nest_svgp_elbo"] = nest_gpr(), nest_svgp()
    for name in nth:
        out[f"nest_gpr_d{name}"] = richardson(lambda x, n=name: nest_gpr(**{n: x}), nth[name])
        out[f"nest_svgp_d{name}"] = richardson(lambda x, n=name: nest_svgp(**{n: x}), nth[name])

    # ---- MAP objective: parameter priors (base.py:201-224, models/model.py:47-76) ---------------------------------------------
    # a prior on the constrained value (Gamma / LogNormal) and one on the UNCONSTRAINED value (Normal + log|Jacobian|); value
    # of log_prior_density / log_posterior_density / training_loss from the reference's statements, and the gradient of the
    # log posterior w.r.t. the unconstrained variables by the same Richardson differences
    import tensorflow_probability as tfp

    def gpr_with_priors():
        mdl = gpflow.models.GPR((Xg, Yg[:, :1]), gpflow.kernels.SquaredExponential(variance=float(th["variance"]),
                                                                                 lengthscales=np.array(th["lengthscales"])),
                                noise_variance=float(th["noise_variance"]))
        mdl.kernel.lengthscales.prior = tfp.distributions.Gamma(np.float64(2.0), np.float64(3.0))
        mdl.kernel.variance.prior = tfp.distributions.LogNormal(np.float64(0.1), np.float64(0.8))
        mdl.likelihood.variance.prior = tfp.distributions.Normal(np.float64(-1.0), np.float64(2.0))
        mdl.likelihood.variance.prior_on = gpflow.base.PriorOn.UNCONSTRAINED
        return mdl
    mp = gpr_with_priors()
    out.update(g_map_log_prior=float(mp.log_prior_density()), g_map_log_posterior=float(mp.log_posterior_density()),
               g_map_training_loss=float(mp.training_loss()))
    for pname, getp in (("lengthscales", lambda mm: mm.kernel.lengthscales), ("variance", lambda mm: mm.kernel.variance),
                        ("noise_variance", lambda mm: mm.likelihood.variance)):
        u0 = _n(getp(mp).unconstrained_variable)
        out[f"g_map_u_{pname}"] = u0

        def f_of_u(u, getp=getp):
            mm = gpr_with_priors()
            par = getp(mm)
            par.assign(_n(par.transform.forward(tf.convert_to_tensor(np.asarray(u, dtype=np.float64)))))
            return float(mm.log_posterior_density())
        out[f"g_map_d{pname}"] = richardson(f_of_u, u0, h=1e-3)

    # ---- natural-gradient parameter conversions (optimizers/natgrad.py:429-516): pure linear algebra, run as they are -----------
    from gpflow.optimizers import natgrad as ref_ng
    rng = np.random.default_rng(71)
    Mn, Pn = 6, 3
    ng_mu = rng.normal(size=(Mn, Pn)); ng_sqrt = np.tril(0.3 * rng.normal(size=(Pn, Mn, Mn))) + 0.7 * np.eye(Mn)
    out.update(ng_mu=ng_mu, ng_sqrt=ng_sqrt)
    n1, n2 = ref_ng.meanvarsqrt_to_natural(T(ng_mu), T(ng_sqrt))
    e1, e2 = ref_ng.meanvarsqrt_to_expectation(T(ng_mu), T(ng_sqrt))
    m_b, s_b = ref_ng.natural_to_meanvarsqrt(n1, n2)
    m_c, s_c = ref_ng.expectation_to_meanvarsqrt(e1, e2)
    out.update(ng_nat1=_n(n1), ng_nat2=_n(n2), ng_eta1=_n(e1), ng_eta2=_n(e2), ng_back_mu=_n(m_b), ng_back_sqrt=_n(s_b),
               ng_back2_mu=_n(m_c), ng_back2_sqrt=_n(s_c))

    # ---- SGPR (models/sgpr.py) ---------------------------------------------------------------------------------------------
    rng = np.random.default_rng(51)
    X = rng.normal(size=(60, 2)); Y = np.sin(X[:, :1]) + 0.1 * rng.normal(size=(60, 1)); Z = X[:9].copy(); Xs = rng.normal(size=(5, 2))
    sg = gpflow.models.SGPR((X, Y), gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.8, 1.2]), Z, noise_variance=0.05)
    mu, var = sg.predict_f(Xs); qmu, qcov = sg.compute_qu()
    out.update(sgpr_X=X, sgpr_Y=Y, sgpr_Z=Z, sgpr_Xnew=Xs, sgpr_elbo=float(sg.elbo()), sgpr_upper=float(sg.upper_bound()),
               sgpr_mu=_n(mu), sgpr_var=_n(var), sgpr_qu_mu=_n(qmu), sgpr_qu_cov=_n(qcov))

    # ---- heteroskedastic Gaussian likelihood (likelihoods/scalar_continuous.py:52-148, utilities/model_utils.py:46-50): the noise
    # scale / variance is a Function of the inputs -- the reference's own recipe of tests/integration/test_linear_noise.py:59
    # (Gaussian(scale=Linear())) with fixed coefficients, and a variance given as a polynomial that dips below the lower bound
    rng = np.random.default_rng(61)
    X = rng.uniform(size=(40, 2)); Y = np.sin(5 * X[:, :1]) + (0.7 - 0.5 * X[:, :1]) * rng.normal(size=(40, 1)); Xs = rng.uniform(size=(6, 2))
    hA, hb = np.array([[-0.5], [0.2]]), np.array([0.7])
    lik = gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=hA, b=hb))
    hk = lambda: gpflow.kernels.SquaredExponential(variance=1.3, lengthscales=[0.3, 0.6])  # noqa: E731
    hg = gpflow.models.GPR((X, Y), hk(), likelihood=lik)
    ymu, yvar = hg.predict_y(Xs); fmu, fvar = hg.predict_f(Xs)
    out.update(het_X=X, het_Y=Y, het_Xnew=Xs, het_A=hA, het_b=hb, het_variance_at=_n(lik.variance_at(X)),
               het_gpr_lml=float(hg.log_marginal_likelihood()), het_gpr_fmu=_n(fmu), het_gpr_fvar=_n(fvar), het_gpr_ymu=_n(ymu),
               het_gpr_yvar=_n(yvar), het_gpr_logdens=_n(hg.predict_log_density((Xs, np.cos(Xs[:, :1])))))
    pmu, pvar = hg.posterior().predict_f(Xs)
    out.update(het_gpr_cached_mu=_n(pmu), het_gpr_cached_var=_n(pvar))
    Zh = X[:7].copy(); hq_mu = 0.3 * rng.normal(size=(7, 1)); hq_sqrt = (np.tril(0.1 * rng.normal(size=(7, 7))) + 0.6 * np.eye(7))[None]
    for wh in (True, False):
        hs = gpflow.models.SVGP(hk(), gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=hA, b=hb)), Zh, q_mu=hq_mu, q_sqrt=hq_sqrt,
                                whiten=wh, num_data=400)
        out[f"het_svgp_elbo_{'white' if wh else 'unwhite'}"] = float(hs.elbo((X, Y)))
        if wh:   # predictions THROUGH the likelihood at new inputs (likelihoods/base.py predict_mean_and_var / predict_log_density)
            symu, syvar = hs.predict_y(Xs)
            out.update(het_svgp_ymu=_n(symu), het_svgp_yvar=_n(syvar), het_svgp_logdens=_n(hs.predict_log_density((Xs, np.cos(Xs[:, :1])))))
    out.update(het_Z=Zh, het_q_mu=hq_mu, het_q_sqrt=hq_sqrt)
    # variance as a Function, clipped at the lower bound where the polynomial goes negative (parameter_or_function.py:52-56)
    pw = np.array([[0.02, 0.3, -0.4, 0.0, 0.0, 0.0]])
    likp = gpflow.likelihoods.Gaussian(variance=gpflow.functions.Polynomial(2, input_dim=2, w=pw), variance_lower_bound=1e-3)
    hp = gpflow.models.GPR((X, Y), hk(), likelihood=likp)
    hsq = gpflow.models.SVGP(hk(), gpflow.likelihoods.Gaussian(variance=gpflow.functions.Polynomial(2, input_dim=2, w=pw), variance_lower_bound=1e-3),
                             Zh, q_mu=hq_mu, q_sqrt=np.abs(0.4 + 0.1 * rng.normal(size=(7, 1))), q_diag=True, num_data=400)
    out.update(het_poly_w=pw, het_poly_variance_at=_n(likp.variance_at(X)), het_poly_gpr_lml=float(hp.log_marginal_likelihood()),
               het_poly_q_sqrt_diag=_n(hsq.q_sqrt), het_poly_svgp_elbo_qdiag=float(hsq.elbo((X, Y))))
    # SGPR under the same likelihood (the reference's tests/integration/test_linear_noise.py recipe, sgpr.py:181-384 with sigma_n per row)
    hsg = gpflow.models.SGPR((X, Y), hk(), Zh, likelihood=gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=hA, b=hb)))
    smu, svar = hsg.predict_f(Xs); squ, sqc = hsg.compute_qu()
    sgymu, sgyvar = hsg.predict_y(Xs)
    out.update(het_sgpr_ymu=_n(sgymu), het_sgpr_yvar=_n(sgyvar), het_sgpr_logdens=_n(hsg.predict_log_density((Xs, np.cos(Xs[:, :1])))))
    out.update(het_sgpr_elbo=float(hsg.elbo()), het_sgpr_mu=_n(smu), het_sgpr_var=_n(svar), het_sgpr_qu_mu=_n(squ), het_sgpr_qu_cov=_n(sqc),
               het_sgpr_upper=float(hsg.upper_bound()))   # sgpr.py:85-148 with sigma_n^2 + c per row (:124-131)
    return out


def main(check: bool) -> None:
    out = build()
    path = os.path.join(HERE, "ref_golden.npz")
    if check:
        ref = np.load(path)
        assert sorted(ref.keys()) == sorted(out.keys()), set(ref.keys()) ^ set(out.keys())
        worst = 0.0
        for k in sorted(out):
            a, b = np.asarray(out[k], dtype=np.float64), ref[k]
            assert a.shape == b.shape, (k, a.shape, b.shape)
            d = float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b))))) if a.size else 0.0
            worst = max(worst, d)
            assert d <= 1e-13, (k, d)
        print(f"ref_golden.npz reproduces from the reference source: {len(out)} arrays, worst scaled difference {worst:.1e}")
    else:
        np.savez_compressed(path, **out)
        print("wrote", path, f"({len(out)} arrays, {os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main("--check" in sys.argv)
