"""CPU: pins the NumPy/SciPy oracle (oracle/gp_oracle.py) against (a) every in-test restatement the
reference's own tests use for this path, (b) the reference's relational tests, (c) the committed golden
vectors.  The reference holds no stored known-answer values for this path."""
import os

import numpy as np
import pytest
import scipy.stats

from oracle import gp_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hotpath_golden.npz")


def ref_rbf_kernel(X, lengthscales, signal_variance):
    """tests/gpflow/kernels/reference.py:13-27 (double loop)."""
    N = X.shape[0]
    K = np.zeros((N, N))
    for i in range(N):
        for j in range(N):
            K[i, j] = signal_variance * np.exp(-0.5 * np.sum(np.square((X[i] - X[j]) / lengthscales)))
    return K


def test_rbf_1d_vs_loop_reference():
    """tests/gpflow/kernels/test_kernels.py:93-101 (ls=1.4, var=2.3)."""
    rng = np.random.RandomState(1)
    X = rng.randn(3, 1)
    np.testing.assert_allclose(orc.rbf_K(X, variance=2.3, lengthscales=1.4), ref_rbf_kernel(X, 1.4, 2.3), rtol=1e-7)
    X = rng.randn(17, 4)
    ls = np.array([0.5, 1.0, 2.0, 3.0])
    np.testing.assert_allclose(orc.rbf_K(X, variance=0.7, lengthscales=ls), ref_rbf_kernel(X, ls, 0.7), rtol=1e-7, atol=1e-15)


@pytest.mark.parametrize("name", ["SquaredExponential", "Matern12", "Matern32", "Matern52"])
def test_kernel_symmetry_diag_and_finite(name):
    """test_kernels.py:259-266,320-327 and test_scaled_euclid_dist.py:40-57 (D=100 negative-r2 hazard)."""
    rng = np.random.RandomState(2)
    X = rng.randn(30, 100)
    K = orc.stationary_K(name, X, variance=1.5, lengthscales=7.0)
    assert np.all(np.isfinite(K))
    np.testing.assert_allclose(K, K.T, rtol=0, atol=1e-14)
    np.testing.assert_allclose(np.diag(K), orc.stationary_K_diag(X, variance=1.5), rtol=1e-7)
    Kc = orc.stationary_K(name, X, X[:7], variance=1.5, lengthscales=7.0)
    np.testing.assert_allclose(Kc, K[:, :7], atol=1e-13)


def test_kuu_kuf_and_schur():
    """tests/gpflow/covariances/test_base_covariances.py:41-47, 99-109."""
    rng = np.random.RandomState(3)
    Z, X = rng.randn(10, 2), rng.randn(25, 2)
    kw = dict(variance=1.1, lengthscales=0.8)
    np.testing.assert_allclose(orc.Kuu(Z, jitter=0.0, **kw), orc.rbf_K(Z, **kw))
    Kuu = orc.Kuu(Z, jitter=1e-6, **kw)
    Kuf = orc.Kuf(Z, X, **kw)
    schur = orc.rbf_K(X, **kw) - Kuf.T @ np.linalg.solve(Kuu, Kuf)
    assert np.all(np.linalg.eigvalsh(schur + 1e-9 * np.eye(25)) > 0)


def test_multivariate_normal_vs_scipy():
    """tests/gpflow/test_logdensities.py:113-129 (4x4)."""
    rng = np.random.RandomState(4)
    A = rng.randn(4, 4); cov = A @ A.T + np.eye(4)
    x, mu = rng.randn(4, 10), rng.randn(4, 1)
    L = np.linalg.cholesky(cov)
    ours = orc.multivariate_normal(x, mu, L)
    ref = np.array([scipy.stats.multivariate_normal.logpdf(x[:, i], mu[:, 0], cov) for i in range(10)])
    np.testing.assert_allclose(ours, ref, rtol=1e-12)


def test_gpr_lml_vs_scipy():
    rng = np.random.default_rng(5)
    X = rng.normal(size=(200, 2)); Y = rng.normal(size=(200, 3))
    kw = dict(variance=1.3, lengthscales=0.7, noise_variance=0.05)
    K = orc.rbf_K(X, variance=1.3, lengthscales=0.7) + 0.05 * np.eye(200)
    ref = sum(scipy.stats.multivariate_normal.logpdf(Y[:, p], np.zeros(200), K) for p in range(3))
    np.testing.assert_allclose(orc.gpr_log_marginal_likelihood(X, Y, **kw), ref, rtol=1e-12)


def test_base_conditional_vs_explicit_inverse():
    """tests/gpflow/conditionals/test_conditionals.py:166-214."""
    rng = np.random.RandomState(0)
    Dy, N, M, Dx = 5, 4, 3, 2
    X, Z = rng.randn(N, Dx), rng.randn(M, Dx)
    q_mu, q_sqrt = rng.randn(M, Dy), np.tril(rng.randn(Dy, M, M))
    Kmm = orc.stationary_K("Matern52", Z, variance=1.0, lengthscales=0.5) + 1e-6 * np.eye(M)
    Kmn = orc.stationary_K("Matern52", Z, X, variance=1.0, lengthscales=0.5)
    Knn = orc.stationary_K("Matern52", X, variance=1.0, lengthscales=0.5)
    Kinv = np.linalg.inv(Kmm)
    S = np.einsum("rij,rkj->rik", q_sqrt, q_sqrt)
    mean_ref = np.einsum("mn,mk,kr->nr", Kmn, Kinv, q_mu)
    cov_ref = Knn[None] + np.einsum("mn,mk,rkl,lj,jp->rnp", Kmn, Kinv, S - Kmm[None], Kinv, Kmn)
    mean, cov = orc.base_conditional(Kmn, Kmm, Knn, q_mu, full_cov=True, q_sqrt=q_sqrt, white=False)
    np.testing.assert_allclose(mean, mean_ref, atol=1e-8)
    np.testing.assert_allclose(cov, cov_ref, atol=1e-8)
    mean2, var = orc.base_conditional(Kmn, Kmm, np.diag(Knn), q_mu, full_cov=False, q_sqrt=q_sqrt, white=False)
    np.testing.assert_allclose(var, np.stack([np.diag(c) for c in cov_ref], -1), atol=1e-8)


def test_conditional_relations():
    """test_conditionals.py:67-129: diag q_sqrt == diag-embedded; whitened == unwhitened after V = L^-1 mu;
    :132-163 upper triangle ignored."""
    rng = np.random.RandomState(6)
    M, N, R = 6, 9, 2
    Z, X = rng.randn(M, 2), rng.randn(N, 2)
    kw = dict(variance=1.0, lengthscales=1.2)
    Kmm = orc.Kuu(Z, jitter=1e-6, **kw); Kmn = orc.Kuf(Z, X, **kw); Knn = orc.stationary_K_diag(X)
    f = rng.randn(M, R)
    qd = rng.rand(M, R) + 0.2
    qf = np.stack([np.diag(qd[:, r]) for r in range(R)])
    for white in (True, False):
        a = orc.base_conditional(Kmn, Kmm, Knn, f, q_sqrt=qd, white=white)
        b = orc.base_conditional(Kmn, Kmm, Knn, f, q_sqrt=qf, white=white)
        np.testing.assert_allclose(a[0], b[0]); np.testing.assert_allclose(a[1], b[1], atol=1e-12)
    q = np.tril(rng.randn(R, M, M))
    L = np.linalg.cholesky(Kmm)
    a = orc.base_conditional(Kmn, Kmm, Knn, f, q_sqrt=q, white=False)
    b = orc.base_conditional(Kmn, Kmm, Knn, np.linalg.solve(L, f), q_sqrt=np.stack([np.linalg.solve(L, x) for x in q]), white=True)
    np.testing.assert_allclose(a[0], b[0], atol=1e-9); np.testing.assert_allclose(a[1], b[1], atol=1e-9)
    c = orc.base_conditional(Kmn, Kmm, Knn, f, q_sqrt=q + np.triu(rng.randn(R, M, M), 1), white=False)
    np.testing.assert_allclose(a[1], c[1])


def compute_kl_1d(q_mu, q_sigma, p_var=1.0):
    """tests/gpflow/test_kullback_leiblers.py:94-98"""
    p_var = np.ones_like(q_mu) if p_var is None else p_var
    q_var = q_sigma ** 2
    return 0.5 * (q_var / p_var + q_mu ** 2 / p_var - 1 + np.log(p_var / q_var))


def multivariate_prior_KL(meanA, covA, meanB, covB):
    """tests/gpflow/models/test_variational.py:92-120 (solve / slogdet form)."""
    K = meanA.shape[0]
    traceTerm = 0.5 * np.trace(np.linalg.solve(covB, covA))
    delta = meanB - meanA
    mahalanobisTerm = 0.5 * np.dot(delta.T, np.linalg.solve(covB, delta))
    constantTerm = -0.5 * K
    logdet = 0.5 * (np.linalg.slogdet(covB)[1] - np.linalg.slogdet(covA)[1])
    return float(np.squeeze(traceTerm + mahalanobisTerm + constantTerm + logdet))


def test_gauss_kl():
    """test_kullback_leiblers.py:121-229."""
    rng = np.random.RandomState(0)
    M, L = 5, 4
    mu = rng.randn(M, L); sqrt = np.array([np.tril(rng.randn(M, M)) for _ in range(L)])
    sqrt_diag = rng.rand(M, L) + 0.1
    A = rng.randn(M, M); K = A @ A.T + 1e-6 * np.eye(M)
    Kb = np.stack([K + i * np.eye(M) for i in range(L)])
    # K vs K_cholesky
    np.testing.assert_allclose(orc.gauss_kl(mu, sqrt, K), orc.gauss_kl(mu, sqrt, K_cholesky=np.linalg.cholesky(K)))
    # diag vs dense
    dense = np.stack([np.diag(sqrt_diag[:, l]) for l in range(L)])
    for KK in (None, K, Kb):
        np.testing.assert_allclose(orc.gauss_kl(mu, sqrt_diag, KK), orc.gauss_kl(mu, dense, KK), rtol=1e-9)
    # K = I vs white
    np.testing.assert_allclose(orc.gauss_kl(mu, sqrt, np.eye(M)), orc.gauss_kl(mu, sqrt), rtol=1e-12)
    # against the solve/slogdet derivation
    tot = sum(multivariate_prior_KL(mu[:, l:l + 1], sqrt[l] @ sqrt[l].T, np.zeros((M, 1)), K) for l in range(L))
    np.testing.assert_allclose(orc.gauss_kl(mu, sqrt, K), tot, rtol=1e-7)
    tot = sum(multivariate_prior_KL(mu[:, l:l + 1], sqrt[l] @ sqrt[l].T, np.zeros((M, 1)), Kb[l]) for l in range(L))
    np.testing.assert_allclose(orc.gauss_kl(mu, sqrt, Kb), tot, rtol=1e-9)
    # 1-D by hand (test_kullback_leiblers.py:213-229)
    m1, s1 = np.array([[0.3]]), np.array([[[0.7]]])
    np.testing.assert_allclose(orc.gauss_kl(m1, s1), compute_kl_1d(0.3, 0.7), rtol=1e-12)
    np.testing.assert_allclose(orc.gauss_kl(m1, s1, np.array([[2.5]])), compute_kl_1d(0.3, 0.7, 2.5), rtol=1e-12)
    with pytest.raises(ValueError):
        orc.gauss_kl(mu, sqrt, K, K_cholesky=K)


def test_svgp_elbo_relations():
    """tests/gpflow/models/test_svgp.py:60-129 (q_diag == diag-embedded, white and not) and the
    method-equivalence identity GPR LML == SVGP bound at Z = X with the optimal q
    (tests/integration/test_method_equivalence.py:181-223, closed form instead of L-BFGS)."""
    rng = np.random.RandomState(0)
    X = rng.randn(20, 1); Y = rng.randn(20, 2) ** 2; Z = rng.randn(3, 1)
    q_mu = rng.randn(3, 2); qd = rng.rand(3, 2) + 0.3
    qf = np.stack([np.diag(qd[:, p]) for p in range(2)])
    kw = dict(variance=1.0, lengthscales=1.0, noise_variance=1.0)
    for w in (True, False):
        np.testing.assert_allclose(orc.svgp_elbo(X, Y, Z, q_mu, qd, whiten=w, **kw),
                                   orc.svgp_elbo(X, Y, Z, q_mu, qf, whiten=w, **kw), rtol=1e-10)
    N = 20
    X = rng.rand(N, 1) * 2; Y = np.sin(3 * X) + 0.3 * rng.randn(N, 1)
    var, ls, nv = 1.3, 0.6, 0.09
    K = orc.rbf_K(X, variance=var, lengthscales=ls) + 1e-6 * np.eye(N)
    Sigma = np.linalg.inv(np.linalg.inv(K) + np.eye(N) / nv)
    mu = Sigma @ (Y / nv)
    elbo = orc.svgp_elbo(X, Y, X, mu, np.linalg.cholesky(Sigma)[None], variance=var, lengthscales=ls,
                         noise_variance=nv, whiten=False)
    lml = orc.gpr_log_marginal_likelihood(X, Y, variance=var, lengthscales=ls, noise_variance=nv)
    np.testing.assert_allclose(elbo, lml, rtol=1e-5)
    # minibatch scaling (test_svgp.py:145-199): sum over a partition with num_data == full-batch value
    Xf = rng.randn(40, 1); Yf = rng.randn(40, 2)
    full = orc.svgp_elbo_terms(Xf, Yf, Z, q_mu, qf, **kw)
    parts = [orc.svgp_elbo_terms(Xf[i:i + 10], Yf[i:i + 10], Z, q_mu, qf, **kw)[0] for i in range(0, 40, 10)]
    np.testing.assert_allclose(sum(parts), full[0], rtol=1e-12)


def test_posterior_cache_equals_fused():
    """tests/gpflow/models/test_svgp_posterior.py:62-90 on the oracle."""
    rng = np.random.RandomState(7)
    Z, X = rng.randn(8, 2), rng.randn(15, 2)
    q_mu, q_sqrt = rng.randn(8, 2), np.tril(rng.randn(2, 8, 8)) * 0.3 + np.eye(8)
    kw = dict(variance=0.9, lengthscales=1.1)
    for w in (True, False):
        for qs in (q_sqrt, np.abs(q_mu) + 0.1):
            a, Q = orc.svgp_precompute(Z, q_mu, qs, whiten=w, **kw)
            m1, v1 = orc.svgp_predict_with_precompute(a, Q, Z, X, **kw)
            m2, v2 = orc.svgp_predict_f(X, Z, q_mu, qs, whiten=w, **kw)
            np.testing.assert_allclose(m1, m2, atol=1e-7); np.testing.assert_allclose(v1, v2, atol=1e-7)


def test_separate_independent_equals_loop():
    rng = np.random.RandomState(8)
    X, Y, Z = rng.randn(30, 2), rng.randn(30, 2), rng.randn(5, 2)
    q_mu, q_sqrt = rng.randn(5, 2), np.tril(rng.randn(2, 5, 5)) * 0.2 + np.eye(5)
    v, l = [1.0, 0.5], [1.0, 2.0]
    tot = orc.svgp_elbo_separate(X, Y, [Z, Z], q_mu, q_sqrt, variances=v, lengthscales_list=l, noise_variance=0.3)
    parts = sum(orc.svgp_elbo(X, Y[:, p:p + 1], Z, q_mu[:, p:p + 1], q_sqrt[p:p + 1], variance=v[p], lengthscales=l[p],
                              noise_variance=0.3) for p in range(2))
    np.testing.assert_allclose(tot, parts, rtol=1e-12)


def test_bijectors():
    x = np.linspace(-5, 5, 11)
    np.testing.assert_allclose(orc.positive_inverse(orc.positive_forward(x, 1e-6), 1e-6), x, atol=1e-9)
    v = np.arange(1.0, 7.0)
    np.testing.assert_array_equal(orc.fill_triangular(v), np.array([[4, 0, 0], [6, 5, 0], [3, 2, 1.0]]))  # TFP doc example
    np.testing.assert_array_equal(orc.fill_triangular_inverse(orc.fill_triangular(v)), v)


def test_golden_vectors_match_oracle():
    g = np.load(GOLD)
    np.testing.assert_allclose(orc.gpr_log_marginal_likelihood(g["gpr_X"], g["gpr_Y"], variance=1.0, lengthscales=2.0,
                                                               noise_variance=1.0), g["gpr_lml"], rtol=1e-13)
    for w in (0, 1):
        np.testing.assert_allclose(orc.svgp_elbo(g["svgp_X"], g["svgp_Y"], g["svgp_Z"], g["svgp_q_mu"], g["svgp_q_sqrt"],
                                                 variance=1.0, lengthscales=1.0, noise_variance=1.0, whiten=bool(w)),
                                   g[f"svgp_elbo_w{w}"], rtol=1e-12)
    np.testing.assert_allclose(orc.gauss_kl(g["kl_mu"], g["kl_sqrt"]), g["kl_white"], rtol=1e-13)
    np.testing.assert_allclose(orc.gauss_kl(g["kl_mu"], g["kl_sqrt"], g["kl_K"]), g["kl_K_val"], rtol=1e-10)
    np.testing.assert_allclose(orc.gpr_log_marginal_likelihood(g["c1_X"], g["c1_Y"], variance=1.0, lengthscales=1.0,
                                                               noise_variance=0.1), g["c1_lml"], rtol=1e-13)


def test_oracle_vs_torch_cpu_second_opinion():
    """Independent fp64 restatement with torch-CPU linalg (cholesky / solve_triangular / logdet) of the two headline
    quantities -- GPR LML and the whitened SVGP ELBO -- on a mid-size problem ('torch-CPU fp64 as a second
    opinion').  Different code, different BLAS: agreement to 1e-10 relative pins the oracle's arithmetic."""
    import torch
    rng = np.random.default_rng(7)
    N, D, M = 300, 3, 40
    X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, 1))
    ls = np.array([0.9, 1.1, 1.3]); var, noise = 1.7, 0.2
    tX, tY = torch.tensor(X), torch.tensor(Y)

    def k(a, b):
        a, b = a / torch.tensor(ls), b / torch.tensor(ls)
        r2 = (a * a).sum(1, keepdim=True) + (b * b).sum(1)[None, :] - 2 * a @ b.T
        return var * torch.exp(-0.5 * r2)

    # GPR LML (gpr.py:91-107, logdensities.py:139-156)
    K = k(tX, tX) + noise * torch.eye(N, dtype=torch.float64)
    L = torch.linalg.cholesky(K)
    alpha = torch.linalg.solve_triangular(L, tY, upper=False)
    lml_t = float(-0.5 * (alpha ** 2).sum() - 0.5 * N * np.log(2 * np.pi) - torch.log(torch.diagonal(L)).sum())
    lml_o = orc.gpr_log_marginal_likelihood(X, Y, variance=var, lengthscales=ls, noise_variance=noise)
    assert abs(lml_t - lml_o) <= 1e-10 * abs(lml_o)
    # whitened SVGP ELBO (svgp.py:166-181, conditionals/util.py:84-169, kullback_leiblers.py:59-165)
    Z = X[:M] + 0.3 * rng.normal(size=(M, D)); tZ = torch.tensor(Z)
    q_mu = 0.3 * rng.normal(size=(M, 1)); q_sqrt = np.tril(0.1 * rng.normal(size=(M, M))) + 0.5 * np.eye(M)
    tq, tS = torch.tensor(q_mu), torch.tensor(q_sqrt)
    Lm = torch.linalg.cholesky(k(tZ, tZ) + 1e-6 * torch.eye(M, dtype=torch.float64))
    A = torch.linalg.solve_triangular(Lm, k(tZ, tX), upper=False)
    fmean = A.T @ tq
    fvar = var - (A * A).sum(0) + ((tS.T @ A) ** 2).sum(0)
    ve = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(noise) - 0.5 * ((tY[:, 0] - fmean[:, 0]) ** 2 + fvar) / noise
    kl = 0.5 * ((tq ** 2).sum() - M - torch.log(torch.diagonal(tS) ** 2).sum() + (tS ** 2).sum())
    elbo_t = float(ve.sum() * (5000.0 / N) - kl)
    elbo_o = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt[None], variance=var, lengthscales=ls, noise_variance=noise, whiten=True,
                           num_data=5000)
    assert abs(elbo_t - elbo_o) <= 1e-10 * abs(elbo_o)


# ----------------------------------------------------------------------------- SGPR
def _sgpr_data(seed=0, N=100, M=20):
    rng = np.random.RandomState(seed)   # tests/gpflow/models/test_sgpr.py:22-26 (Datum)
    X = rng.randn(N, 2); Y = np.sin(X @ np.array([[-1.4], [0.5]])) + 0.5 * rng.randn(N, 1); Z = rng.randn(M, 2)
    return X, Y, Z, dict(variance=1.3, lengthscales=0.9, noise_variance=0.4)


def test_sgpr_compute_qu_equals_predict_at_Z():
    """tests/gpflow/models/test_sgpr.py:29-44 (test_sgpr_qu), without the optimisation step."""
    X, Y, Z, kw = _sgpr_data()
    # (the two routes differ by O(jitter * cond): Kus = k(Z, Z) carries no jitter, kuu does -- 5e-5 at the default 1e-6
    #  for these un-optimised hyper-parameters, so the identity is checked at a smaller jitter)
    kw = dict(kw, jitter=1e-9)
    mu, cov = orc.sgpr_compute_qu(X, Y, Z, **kw)
    fm, fc = orc.sgpr_predict_f(X, Y, Z, Z, full_cov=True, **kw)
    np.testing.assert_allclose(mu, fm, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cov[None], fc, rtol=1e-5, atol=1e-5)


def test_sgpr_predicts_like_unwhitened_svgp_with_its_qu():
    """tests/gpflow/models/test_sgpr.py:47-80 (test_sgpr_svgp_qu_equivivalent), constant noise."""
    X, Y, Z, kw = _sgpr_data(2)
    mu, cov = orc.sgpr_compute_qu(X, Y, Z, **kw)
    q_sqrt = np.linalg.cholesky(cov)[None]
    Xnew = np.random.RandomState(3).randn(100, 2)
    fm, fv = orc.sgpr_predict_f(X, Y, Z, Xnew, **kw)
    sm, sv = orc.svgp_predict_f(Xnew, Z, mu, q_sqrt, variance=kw["variance"], lengthscales=kw["lengthscales"], whiten=False)
    np.testing.assert_allclose(fm, sm, atol=1e-4)
    np.testing.assert_allclose(fv, sv, atol=1e-4)


def test_sgpr_bounds_sandwich_the_exact_lml_and_are_tight_at_Z_equals_X():
    """elbo <= exact LML <= upper_bound (Titsias 2009 / 2014), with equality of the ELBO when Z = X."""
    X, Y, Z, kw = _sgpr_data(4, N=80, M=15)
    lml = float(orc.gpr_log_marginal_likelihood(X, Y, **kw))
    lo, hi = orc.sgpr_elbo(X, Y, Z, **kw), orc.sgpr_upper_bound(X, Y, Z, **kw)
    assert lo <= lml <= hi
    assert abs(orc.sgpr_elbo(X, Y, X, jitter=1e-10, **kw) - lml) <= 1e-5 * abs(lml)


def test_nextrows_golden_vectors_match_oracles():
    """tests/golden/nextrows_golden.npz (SGPR, gradients, natural gradient; generated by make_golden_next.py on the
    reference's test fixtures) is reproduced by the oracles -- a pin against silent drift of the checkers themselves."""
    from oracle import gp_oracle_grad as orcg
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "nextrows_golden.npz"))
    kw = dict(variance=1.0, lengthscales=1.0, noise_variance=1.0)
    X, Y, Z = g["sgpr_X"], g["sgpr_Y"], g["sgpr_Z"]
    np.testing.assert_allclose(orc.sgpr_elbo(X, Y, Z, **kw), g["sgpr_elbo"], rtol=1e-12)
    np.testing.assert_allclose(orc.sgpr_upper_bound(X, Y, Z, **kw), g["sgpr_upper"], rtol=1e-12)
    fm, fv = orc.sgpr_predict_f(X, Y, Z, g["sgpr_Xnew"], **kw)
    np.testing.assert_allclose(fm, g["sgpr_mean"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(fv, g["sgpr_var"], rtol=0, atol=1e-12)
    v, gr = orcg.svgp_elbo_value_and_grads(g["grad_X"], g["grad_Y"], g["grad_Z"], g["grad_q_mu"], g["grad_q_sqrt"],
                                           num_data=200, **kw)
    np.testing.assert_allclose(v, g["grad_elbo"], rtol=1e-12)
    for k, val in gr.items():
        np.testing.assert_allclose(val, g[f"grad_g_{k}"], rtol=0, atol=1e-10 * max(1.0, np.abs(val).max()))
    mu_n, sq_n = orcg.natgrad_step(g["grad_q_mu"], g["grad_q_sqrt"], -gr["q_mu"], -gr["q_sqrt"], 0.3)
    np.testing.assert_allclose(mu_n, g["nat_q_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(sq_n, g["nat_q_sqrt"], rtol=0, atol=1e-10)
