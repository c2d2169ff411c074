"""CPU oracle: NumPy/SciPy fp64 restatement of GPflow's dense-GP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gpflow_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
there only as the checker / timed CPU baseline -- never as the thing shipped.

Parity status: PINNED to the reference's own source.  The reference (GPflow 2.9.2) does its arithmetic in TensorFlow,
which is not installable in the build container, and its tests hold no stored golden vectors for this path; so
tests/golden/refshim/ provides NumPy stand-ins for exactly the third-party surface GPflow touches, the UNMODIFIED package
under /root/reference is imported over them, and tests/golden/make_golden_ref.py writes what its public API returns --
kernels, GPR / SVGP / SGPR objectives and predictions, conditionals, all seven gauss_kl forms, multi-output posteriors,
and (round 4) finite-difference gradients of the three objectives -- to tests/golden/ref_golden.npz.  This oracle
reproduces every one of those arrays at 1e-12 and its autograd twin (gp_oracle_grad.py) the gradients at 1e-7
(tests/test_reference_golden.py).  It is additionally pinned by (tests/test_oracle.py):
  * every in-test restatement the reference's own tests use (loop RBF kernel
    tests/gpflow/kernels/reference.py:13-27, scipy mvn.logpdf tests/gpflow/test_logdensities.py:113-129,
    explicit-inverse conditional tests/gpflow/conditionals/test_conditionals.py:166-214,
    by-hand 1-D KL tests/gpflow/test_kullback_leiblers.py:94-98,213-229, slogdet KL
    tests/gpflow/models/test_variational.py:92-120),
  * the reference's relational tests (q_diag == diag-embedded q_sqrt, whitened == unwhitened
    after V = L^-1 mu, K vs K_cholesky, GPR LML == SVGP ELBO at Z = X with the optimal q).
What stays unpinned is TensorFlow's own rounding (Eigen LLT against LAPACK dpotrf: both IEEE fp64) -- stated in DESIGN.md.

Each function cites the reference file:line it restates (paths relative to the GPflow tree).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

DEFAULT_JITTER = 1e-6  # gpflow/config/__config__.py:104
LOG2PI = np.log(2 * np.pi)


# ----------------------------------------------------------------------------- L2: covariances
def square_distance(X, X2=None):
    """gpflow/utilities/ops.py:105-122 -- the *expansion* formula (can go slightly negative)."""
    X = np.asarray(X, dtype=np.float64)
    if X2 is None:
        Xs = np.sum(np.square(X), axis=-1, keepdims=True)
        dist = -2 * (X @ X.T)
        dist += Xs + Xs.T
        return dist
    X2 = np.asarray(X2, dtype=np.float64)
    Xs = np.sum(np.square(X), axis=-1)
    X2s = np.sum(np.square(X2), axis=-1)
    dist = -2 * np.tensordot(X, X2, [[-1], [-1]])
    dist += Xs[:, None] + X2s[None, :]
    return dist


def scale(X, lengthscales):
    """gpflow/kernels/stationaries.py:77-79"""
    return None if X is None else np.asarray(X, dtype=np.float64) / lengthscales


def scaled_squared_euclid_dist(X, X2, lengthscales):
    """gpflow/kernels/stationaries.py:124-130"""
    return square_distance(scale(X, lengthscales), scale(X2, lengthscales))


def rbf_K(X, X2=None, *, variance=1.0, lengthscales=1.0):
    """SquaredExponential.K: stationaries.py:103-105 + K_r2 :209-210"""
    r2 = scaled_squared_euclid_dist(X, X2, lengthscales)
    return variance * np.exp(-0.5 * r2)


def stationary_K(name, X, X2=None, *, variance=1.0, lengthscales=1.0):
    """SE (:209-210), Matern12 (:254-255), Matern32 (:281-283), Matern52 (:311-313), with the
    r = sqrt(max(r2, 1e-36)) clamp of IsotropicStationary.K_r2 (:111-116)."""
    r2 = scaled_squared_euclid_dist(X, X2, lengthscales)
    if name in ("SquaredExponential", "RBF"):
        return variance * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, 1e-36))
    if name == "Matern12":
        return variance * np.exp(-r)
    if name == "Matern32":
        sqrt3 = np.sqrt(3.0)
        return variance * (1.0 + sqrt3 * r) * np.exp(-sqrt3 * r)
    if name == "Matern52":
        sqrt5 = np.sqrt(5.0)
        return variance * (1.0 + sqrt5 * r + 5.0 / 3.0 * np.square(r)) * np.exp(-sqrt5 * r)
    raise ValueError(name)


def stationary_K_diag(X, *, variance=1.0):
    """Stationary.K_diag: stationaries.py:82-83 -- exactly sigma^2."""
    return np.full(np.asarray(X).shape[:-1], float(variance))


def add_noise_cov(K, likelihood_variance):
    """gpflow/utilities/model_utils.py:33-38 (set_diag(K, diag_part(K) + s))"""
    K = np.array(K, dtype=np.float64, copy=True)
    idx = np.arange(K.shape[-1])
    K[..., idx, idx] = K[..., idx, idx] + likelihood_variance
    return K


def Kuu(Z, *, variance, lengthscales, jitter=0.0, kernel="SquaredExponential"):
    """gpflow/covariances/kuus.py:24-34"""
    Kzz = stationary_K(kernel, Z, None, variance=variance, lengthscales=lengthscales)
    Kzz += jitter * np.eye(Kzz.shape[0])
    return Kzz


def Kuf(Z, Xnew, *, variance, lengthscales, kernel="SquaredExponential"):
    """gpflow/covariances/kufs.py:25-34"""
    return stationary_K(kernel, Z, Xnew, variance=variance, lengthscales=lengthscales)


# ----------------------------------------------------------------------------- L3: log densities
def gaussian_logdensity(x, mu, var):
    """gpflow/logdensities.py:29-30"""
    return -0.5 * (LOG2PI + np.log(var) + np.square(mu - x) / var)


def multivariate_normal(x, mu, L):
    """gpflow/logdensities.py:139-156"""
    d = x - mu
    alpha = sla.solve_triangular(L, d, lower=True)
    num_dims = d.shape[0]
    p = -0.5 * np.sum(np.square(alpha), 0)
    p -= 0.5 * num_dims * LOG2PI
    p -= np.sum(np.log(np.diag(L)))
    return p


# ----------------------------------------------------------------------------- L3: conditionals
def base_conditional(Kmn, Kmm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """gpflow/conditionals/util.py:37-70"""
    Lm = np.linalg.cholesky(Kmm)
    return base_conditional_with_lm(Kmn, Lm, Knn, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white)


def base_conditional_with_lm(Kmn, Lm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """gpflow/conditionals/util.py:84-169 (Kmn [M,N] only; leading batch dims are flattened
    by the caller exactly as the reference's transpose :108-119 does)."""
    num_func = f.shape[-1]
    A = sla.solve_triangular(Lm, Kmn, lower=True)  # :125
    if full_cov:
        fvar = Knn - A.T @ A  # :129
        fvar = np.broadcast_to(fvar[None], (num_func,) + fvar.shape).copy()
    else:
        fvar = Knn - np.sum(np.square(A), -2)  # :133
        fvar = np.broadcast_to(fvar[None], (num_func,) + fvar.shape).copy()  # [R, N]
    if not white:
        A = sla.solve_triangular(Lm.T, A, lower=False)  # :139
    fmean = A.T @ f  # :144
    if q_sqrt is not None:
        if q_sqrt.ndim == 2:
            LTA = A * q_sqrt.T[:, :, None]  # :149  [R, M, N]
        elif q_sqrt.ndim == 3:
            L = np.tril(q_sqrt)  # :151 band_part(q_sqrt, -1, 0)
            LTA = np.einsum("rkm,kn->rmn", L, A)  # :157  L^T A per latent
        else:
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))
        if full_cov:
            fvar = fvar + np.einsum("rmn,rmk->rnk", LTA, LTA)  # :162
        else:
            fvar = fvar + np.sum(np.square(LTA), -2)  # :164
    if not full_cov:
        fvar = fvar.T  # :167  [N, R]
    return fmean, fvar


def expand_independent_outputs(fvar, full_cov, full_output_cov):
    """gpflow/conditionals/util.py:222-239"""
    if full_cov and full_output_cov:
        P, N, _ = fvar.shape
        out = np.zeros((N, P, N, P))
        for p in range(P):
            out[:, p, :, p] = fvar[p]
        return out
    if (not full_cov) and full_output_cov:
        N, P = fvar.shape
        out = np.zeros((N, P, P))
        idx = np.arange(P)
        out[:, idx, idx] = fvar
        return out
    return fvar


def separate_independent_conditional(Kmns, Kmms, Knns, f, *, full_cov=False, q_sqrt=None, white=False):
    """gpflow/conditionals/util.py:566-629 -- the map_fn loop over P independent GPs."""
    P = Kmms.shape[0]
    mus, vs = [], []
    for p in range(P):
        qs = None
        if q_sqrt is not None:
            qs = q_sqrt[:, p : p + 1] if q_sqrt.ndim == 2 else q_sqrt[p : p + 1]
        mu, var = base_conditional(
            Kmns[p], Kmms[p], Knns[p], f[:, p : p + 1], full_cov=full_cov, q_sqrt=qs, white=white
        )
        mus.append(mu[:, 0])
        vs.append(var[0] if full_cov else var[:, 0])
    fmu = np.stack(mus, axis=-1)
    fvar = np.stack(vs, axis=0) if full_cov else np.stack(vs, axis=-1)
    return fmu, fvar


# ----------------------------------------------------------------------------- L3: KL
def gauss_kl(q_mu, q_sqrt, K=None, *, K_cholesky=None):
    """gpflow/kullback_leiblers.py:59-165"""
    if (K is not None) and (K_cholesky is not None):
        raise ValueError(
            "Ambiguous arguments: gauss_kl() must only be passed one of `K` or `K_cholesky`."
        )
    is_white = (K is None) and (K_cholesky is None)
    is_diag = q_sqrt.ndim == 2
    M, L = q_mu.shape
    is_batched = False
    if is_white:
        alpha = q_mu
    else:
        Lp = np.linalg.cholesky(K) if K is not None else K_cholesky
        is_batched = Lp.ndim == 3
        if is_batched:
            alpha = np.stack(
                [sla.solve_triangular(Lp[l], q_mu[:, l], lower=True) for l in range(L)], axis=0
            )
        else:
            alpha = sla.solve_triangular(Lp, q_mu, lower=True)
    if is_diag:
        Lq = Lq_diag = q_sqrt
        Lq_full = np.stack([np.diag(q_sqrt[:, l]) for l in range(L)], axis=0)
    else:
        Lq = Lq_full = np.tril(q_sqrt)
        Lq_diag = np.stack([np.diag(Lq[l]) for l in range(L)], axis=-1)
    mahalanobis = np.sum(np.square(alpha))
    constant = -float(q_mu.size)
    logdet_qcov = np.sum(np.log(np.square(Lq_diag)))
    if is_white:
        trace = np.sum(np.square(Lq))
    else:
        if is_diag and not is_batched:
            Lp_inv = sla.solve_triangular(Lp, np.eye(M), lower=True)
            K_inv = np.diag(sla.solve_triangular(Lp.T, Lp_inv, lower=False))[:, None]
            trace = np.sum(K_inv * np.square(q_sqrt))
        else:
            tr = 0.0
            for l in range(L):
                Lp_l = Lp[l] if is_batched else Lp
                LpiLq = sla.solve_triangular(Lp_l, Lq_full[l], lower=True)
                tr += np.sum(np.square(LpiLq))
            trace = tr
    twoKL = mahalanobis + constant - logdet_qcov + trace
    if not is_white:
        if is_batched:
            diag = np.stack([np.diag(Lp[l]) for l in range(L)])
        else:
            diag = np.diag(Lp)
        sum_log_sqdiag_Lp = np.sum(np.log(np.square(diag)))
        scale_ = 1.0 if is_batched else float(L)
        twoKL += scale_ * sum_log_sqdiag_Lp
    return 0.5 * twoKL


def prior_kl(Z, q_mu, q_sqrt, *, variance, lengthscales, whiten=False, kernel="SquaredExponential"):
    """gpflow/kullback_leiblers.py:31-49 (InducingPoints x stationary kernel)."""
    if whiten:
        return gauss_kl(q_mu, q_sqrt, None)
    K = Kuu(Z, variance=variance, lengthscales=lengthscales, jitter=DEFAULT_JITTER, kernel=kernel)
    return gauss_kl(q_mu, q_sqrt, K)


# ----------------------------------------------------------------------------- L3: likelihood
def _noise_column(noise_variance):
    """A constant, or one variance per data row [N] -> [N, 1] (Gaussian._variance for a Function-valued variance / scale:
    likelihoods/scalar_continuous.py:92-105 returns [N, 1], which then broadcasts against [N, P])."""
    nv = np.asarray(noise_variance, dtype=np.float64)
    return nv[:, None] if nv.ndim == 1 else nv


def gaussian_variance_at(X, *, variance=None, scale=None, lower_bound=1e-6):
    """Gaussian._variance (likelihoods/scalar_continuous.py:92-105) for a heteroskedastic likelihood: `variance` / `scale` is a
    callable of X returning [N, 1]; its value is clipped from below at the lower bound (of the variance, or its square root for
    a scale) when evaluated (utilities/parameter_or_function.py:45-58).  Returns one variance per row [N]."""
    if variance is not None:
        v = np.maximum(np.asarray(variance(X), dtype=np.float64), lower_bound)
    else:
        v = np.maximum(np.asarray(scale(X), dtype=np.float64), np.sqrt(lower_bound)) ** 2
    return np.broadcast_to(v, (np.asarray(X).shape[0], 1))[:, 0].copy()


def linear_function(A, b):
    """gpflow/functions.py:96-126: X -> X A + b"""
    A, b = np.atleast_2d(np.asarray(A, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
    return lambda X: np.tensordot(np.asarray(X, dtype=np.float64), A, axes=([-1], [0])) + b


def gaussian_variational_expectations(Fmu, Fvar, Y, noise_variance):
    """gpflow/likelihoods/scalar_continuous.py:139-148 (noise_variance: a constant or one value per row [N])"""
    noise_variance = _noise_column(noise_variance)
    return np.sum(
        -0.5 * LOG2PI - 0.5 * np.log(noise_variance) - 0.5 * ((Y - Fmu) ** 2 + Fvar) / noise_variance,
        axis=-1,
    )


def gaussian_predict_mean_and_var(Fmu, Fvar, noise_variance):
    """scalar_continuous.py:127-130"""
    return Fmu.copy(), Fvar + _noise_column(noise_variance)


def gaussian_predict_log_density(Fmu, Fvar, Y, noise_variance):
    """scalar_continuous.py:132-136"""
    return np.sum(gaussian_logdensity(Y, Fmu, Fvar + _noise_column(noise_variance)), axis=-1)


# ----------------------------------------------------------------------------- L4: GPR
def gpr_log_marginal_likelihood(X, Y, *, variance, lengthscales, noise_variance, mean=0.0,
                                kernel="SquaredExponential"):
    """gpflow/models/gpr.py:91-107 (noise_variance: a constant, or likelihood.variance_at(X) squeezed to [N] -- add_likelihood_noise_cov,
    utilities/model_utils.py:46-50)"""
    K = stationary_K(kernel, X, None, variance=variance, lengthscales=lengthscales)
    ks = add_noise_cov(K, noise_variance)
    L = np.linalg.cholesky(ks)
    m = np.full_like(Y, mean)
    log_prob = multivariate_normal(Y, m, L)
    return np.sum(log_prob)


def gpr_predict_f(X, Y, Xnew, *, variance, lengthscales, noise_variance, full_cov=False, mean=0.0,
                  kernel="SquaredExponential"):
    """gpflow/models/gpr.py:178-190 -> posteriors.py:384-443 (GPRPosterior fused path)."""
    err = Y - mean
    Kmm = stationary_K(kernel, X, None, variance=variance, lengthscales=lengthscales)
    Lm = np.linalg.cholesky(add_noise_cov(Kmm, noise_variance))
    if full_cov:
        Knn = stationary_K(kernel, Xnew, None, variance=variance, lengthscales=lengthscales)
    else:
        Knn = stationary_K_diag(Xnew, variance=variance)
    Kmn = stationary_K(kernel, X, Xnew, variance=variance, lengthscales=lengthscales)
    fmean, fvar = base_conditional_with_lm(Kmn, Lm, Knn, err, full_cov=full_cov, q_sqrt=None, white=False)
    return fmean + mean, fvar


# ----------------------------------------------------------------------------- L4: SVGP
def svgp_predict_f(Xnew, Z, q_mu, q_sqrt, *, variance, lengthscales, whiten=True, full_cov=False,
                   mean=0.0, kernel="SquaredExponential"):
    """gpflow/models/svgp.py:243-255 -> posteriors.py:828-841 (IndependentPosteriorSingleOutput)
    and the SharedIndependent/SharedIV branch posteriors.py:849-861 (identical arithmetic)."""
    if full_cov:
        Knn = stationary_K(kernel, Xnew, None, variance=variance, lengthscales=lengthscales)
    else:
        Knn = stationary_K_diag(Xnew, variance=variance)
    Kmm = Kuu(Z, variance=variance, lengthscales=lengthscales, jitter=DEFAULT_JITTER, kernel=kernel)
    Kmn = Kuf(Z, Xnew, variance=variance, lengthscales=lengthscales, kernel=kernel)
    fmean, fvar = base_conditional(Kmn, Kmm, Knn, q_mu, full_cov=full_cov, q_sqrt=q_sqrt, white=whiten)
    return fmean + mean, fvar


def svgp_elbo_terms(X, Y, Z, q_mu, q_sqrt, *, variance, lengthscales, noise_variance, whiten=True,
                    mean=0.0, kernel="SquaredExponential"):
    """The two pieces SVGP.elbo combines (svgp.py:172-174): (sum var_exp over rows, KL)."""
    kl = prior_kl(Z, q_mu, q_sqrt, variance=variance, lengthscales=lengthscales, whiten=whiten,
                  kernel=kernel)
    f_mean, f_var = svgp_predict_f(X, Z, q_mu, q_sqrt, variance=variance, lengthscales=lengthscales,
                                   whiten=whiten, mean=mean, kernel=kernel)
    var_exp = gaussian_variational_expectations(f_mean, f_var, Y, noise_variance)
    return np.sum(var_exp), kl


def svgp_elbo(X, Y, Z, q_mu, q_sqrt, *, variance, lengthscales, noise_variance, whiten=True,
              num_data=None, mean=0.0, kernel="SquaredExponential"):
    """gpflow/models/svgp.py:166-181"""
    s, kl = svgp_elbo_terms(X, Y, Z, q_mu, q_sqrt, variance=variance, lengthscales=lengthscales,
                            noise_variance=noise_variance, whiten=whiten, mean=mean, kernel=kernel)
    scale_ = 1.0 if num_data is None else float(num_data) / X.shape[0]
    return s * scale_ - kl


def svgp_elbo_separate(X, Y, Zs, q_mu, q_sqrt, *, variances, lengthscales_list, noise_variance,
                       whiten=True, num_data=None):
    """SeparateIndependent kernels + (Shared|Separate) inducing variables:
    posteriors.py:862-887 -> conditionals/util.py:566-629; KL with K [L,M,M] kullback_leiblers.py:48-49."""
    P = len(variances)
    Kmms = np.stack([Kuu(Zs[p], variance=variances[p], lengthscales=lengthscales_list[p],
                         jitter=DEFAULT_JITTER) for p in range(P)])
    Kmns = np.stack([Kuf(Zs[p], X, variance=variances[p], lengthscales=lengthscales_list[p])
                     for p in range(P)])
    Knns = np.stack([stationary_K_diag(X, variance=variances[p]) for p in range(P)])
    fmu, fvar = separate_independent_conditional(Kmns, Kmms, Knns, q_mu, q_sqrt=q_sqrt, white=whiten)
    var_exp = gaussian_variational_expectations(fmu, fvar, Y, noise_variance)
    kl = gauss_kl(q_mu, q_sqrt, None) if whiten else gauss_kl(q_mu, q_sqrt, Kmms)
    scale_ = 1.0 if num_data is None else float(num_data) / X.shape[0]
    return np.sum(var_exp) * scale_ - kl


# ----------------------------------------------------------------------------- SGPR
def _noise_rows(noise_variance, N):
    """sigma_n^2 per data row [N]: a constant broadcast, or likelihood.variance_at(X) squeezed (sgpr.py:207)"""
    return np.broadcast_to(np.asarray(noise_variance, dtype=np.float64).reshape(-1), (N,)).copy()


def sgpr_common(X, Z, *, variance, lengthscales, noise_variance, jitter=DEFAULT_JITTER):
    """gpflow/models/sgpr.py:181-213 (_common_calculation); noise_variance a constant or one value per data row [N]."""
    sigma = np.sqrt(_noise_rows(noise_variance, np.asarray(X).shape[0]))
    kuf = Kuf(Z, X, variance=variance, lengthscales=lengthscales)
    kuu = Kuu(Z, variance=variance, lengthscales=lengthscales, jitter=jitter)
    L = np.linalg.cholesky(kuu)
    A = sla.solve_triangular(L, kuf / sigma, lower=True)
    AAT = A @ A.T
    B = AAT + np.eye(AAT.shape[0])
    LB = np.linalg.cholesky(B)
    return A, AAT, LB, L


def sgpr_elbo(X, Y, Z, *, variance, lengthscales, noise_variance, mean=0.0, jitter=DEFAULT_JITTER):
    """gpflow/models/sgpr.py:214-290 (logdet_term, quad_term, elbo)."""
    N, P = Y.shape
    A, AAT, LB, _ = sgpr_common(X, Z, variance=variance, lengthscales=lengthscales, noise_variance=noise_variance,
                                jitter=jitter)
    nv = _noise_rows(noise_variance, N)
    trace_k = np.sum(variance / nv)                              # :236-238  (K_diag = variance)
    trace_q = np.trace(AAT)                                      # :240
    half_logdet_b = np.sum(np.log(np.diag(LB)))                  # :245
    logdet = -P * (half_logdet_b + 0.5 * np.sum(np.log(nv)) + 0.5 * (trace_k - trace_q))   # :248-251
    err = (Y - mean) / np.sqrt(nv)[:, None]                      # :266
    c = sla.solve_triangular(LB, A @ err, lower=True)            # :268-269
    quad = -0.5 * (np.sum(err * err) - np.sum(c * c))            # :272-276
    const = -0.5 * N * P * LOG2PI                                # :287
    return const + logdet + quad


def sgpr_predict_f(X, Y, Z, Xnew, *, variance, lengthscales, noise_variance, mean=0.0, full_cov=False,
                   jitter=DEFAULT_JITTER):
    """gpflow/models/sgpr.py:292-345"""
    sigma = np.sqrt(_noise_rows(noise_variance, np.asarray(X).shape[0]))[:, None]
    A, _, LB, L = sgpr_common(X, Z, variance=variance, lengthscales=lengthscales, noise_variance=noise_variance,
                              jitter=jitter)
    Kus = Kuf(Z, Xnew, variance=variance, lengthscales=lengthscales)
    c = sla.solve_triangular(LB, A @ ((Y - mean) / sigma), lower=True)
    tmp1 = sla.solve_triangular(L, Kus, lower=True)
    tmp2 = sla.solve_triangular(LB, tmp1, lower=True)
    fmean = tmp2.T @ c + mean
    P = Y.shape[1]
    if full_cov:
        var = rbf_K(Xnew, variance=variance, lengthscales=lengthscales) + tmp2.T @ tmp2 - tmp1.T @ tmp1
        return fmean, np.tile(var[None], [P, 1, 1])
    var = variance + np.sum(tmp2 * tmp2, 0) - np.sum(tmp1 * tmp1, 0)
    return fmean, np.tile(var[:, None], [1, P])


def sgpr_compute_qu(X, Y, Z, *, variance, lengthscales, noise_variance, mean=0.0, jitter=DEFAULT_JITTER):
    """gpflow/models/sgpr.py:351-384: q(u) = N(mu, cov)."""
    std = np.sqrt(_noise_rows(noise_variance, np.asarray(X).shape[0]))
    kuf = Kuf(Z, X, variance=variance, lengthscales=lengthscales)
    kuu = Kuu(Z, variance=variance, lengthscales=lengthscales, jitter=jitter)
    skuf = kuf / std
    sig_sqrt = np.linalg.cholesky(kuu + skuf @ skuf.T)
    sig_sqrt_kuu = sla.solve_triangular(sig_sqrt, kuu, lower=True)
    cov = sig_sqrt_kuu.T @ sig_sqrt_kuu
    mu = sig_sqrt_kuu.T @ sla.solve_triangular(sig_sqrt, skuf @ ((Y - mean) / std[:, None]), lower=True)
    return mu, cov


def sgpr_upper_bound(X, Y, Z, *, variance, lengthscales, noise_variance, mean=0.0, jitter=DEFAULT_JITTER):
    """gpflow/models/sgpr.py:85-148 (Titsias 2014 upper bound), P = 1 semantics as written; noise_variance a constant or one
    value per data row [N] (likelihood.variance_at(X), :108): every row is then rescaled by ITS sigma_n^2 + c (:124-131)."""
    N = X.shape[0]
    sigma_sq = _noise_rows(noise_variance, N)                                   # :108
    kuu = Kuu(Z, variance=variance, lengthscales=lengthscales, jitter=jitter)
    kuf = Kuf(Z, X, variance=variance, lengthscales=lengthscales)
    L = np.linalg.cholesky(kuu)
    A = sla.solve_triangular(L, kuf, lower=True)                                # :118
    A_sigma = sla.solve_triangular(L, kuf / np.sqrt(sigma_sq), lower=True)      # :120
    LB = np.linalg.cholesky(np.eye(len(Z)) + A_sigma @ A_sigma.T)               # :121-123
    c = N * variance - np.sum(A * A)                                            # :126  (K_diag = variance)
    cn_var = sigma_sq + c                                                       # :129
    cn_std = np.sqrt(cn_var)
    const = -0.5 * np.sum(np.log(2 * np.pi * sigma_sq))                         # :132
    logdet = -np.sum(np.log(np.diag(LB)))                                       # :133
    A_cn = sla.solve_triangular(L, kuf / cn_std, lower=True)                    # :135
    err = Y - mean
    LC = np.linalg.cholesky(np.eye(len(Z)) + A_cn @ A_cn.T)                     # :139
    v = sla.solve_triangular(LC, A_cn @ (err / cn_std[:, None]), lower=True)    # :140-142
    quad = -0.5 * np.sum((err / cn_std[:, None]) ** 2) + 0.5 * np.sum(v * v)    # :143-145
    return const + logdet + quad


# ----------------------------------------------------------------------------- posteriors cache
def svgp_precompute(Z, q_mu, q_sqrt, *, variance, lengthscales, whiten=True,
                    kernel="SquaredExponential"):
    """gpflow/posteriors.py:694-746 (BasePosterior._precompute, [M,M] Kuu): alpha [M,L], Qinv [L,M,M]."""
    Kuu_ = Kuu(Z, variance=variance, lengthscales=lengthscales, jitter=DEFAULT_JITTER, kernel=kernel)
    M, Lnum = q_mu.shape
    L = np.linalg.cholesky(Kuu_)
    if not whiten:
        alpha = sla.cho_solve((L, True), q_mu)
    else:
        alpha = sla.solve_triangular(L.T, q_mu, lower=False)
    I = np.eye(M)
    if q_sqrt is None:
        B = np.broadcast_to(I, (1, M, M))
    else:
        if q_sqrt.ndim == 2:
            qs = np.stack([np.diag(q_sqrt[:, l]) for l in range(Lnum)])
        else:
            qs = q_sqrt
        if not whiten:
            Linv_qsqrt = np.stack([sla.solve_triangular(L, qs[l], lower=True) for l in range(Lnum)])
            cov = np.einsum("lij,lkj->lik", Linv_qsqrt, Linv_qsqrt)
        else:
            cov = np.einsum("lij,lkj->lik", qs, qs)
        B = I[None] - cov
    Qinv = []
    for b in B:
        LinvT_B = sla.solve_triangular(L.T, b, lower=False)
        B_Linv = LinvT_B.T
        Qinv.append(sla.solve_triangular(L.T, B_Linv, lower=False))
    Qinv = np.broadcast_to(np.stack(Qinv), (Lnum, M, M)).copy()
    return alpha, Qinv


def svgp_predict_with_precompute(alpha, Qinv, Z, Xnew, *, variance, lengthscales, full_cov=False,
                                 mean=0.0, kernel="SquaredExponential"):
    """gpflow/posteriors.py:794-822"""
    Kuf_ = Kuf(Z, Xnew, variance=variance, lengthscales=lengthscales, kernel=kernel)
    mean_ = Kuf_.T @ alpha
    if full_cov:
        Kff = stationary_K(kernel, Xnew, None, variance=variance, lengthscales=lengthscales)
        cov = Kff[None] - np.einsum("mn,lmk,kj->lnj", Kuf_, Qinv, Kuf_)
    else:
        Kff = stationary_K_diag(Xnew, variance=variance)
        cov = Kff[None] - np.sum(Kuf_[None] * (Qinv @ Kuf_), axis=-2)
        cov = cov.T
    return mean_ + mean, cov


# ----------------------------------------------------------------------------- L1: bijectors
def softplus(x):
    return np.logaddexp(0.0, x)


def softplus_inverse(y):
    """tfp.math.softplus_inverse: log(expm1(y)), stable form."""
    y = np.asarray(y, dtype=np.float64)
    return y + np.log(-np.expm1(-y))


def positive_forward(x, lower=0.0):
    """gpflow/utilities/bijectors.py:27-45: Chain([Shift(lower), Softplus()])"""
    return softplus(x) + lower


def positive_inverse(y, lower=0.0):
    return softplus_inverse(np.asarray(y, dtype=np.float64) - lower)


def fill_triangular(x):
    """tfp.bijectors.FillTriangular (tfp.math.fill_triangular, lower): vector [n(n+1)/2] -> [n,n].
    Restated from TFP's documented algorithm: concat(x[..., n:], reverse(x)) reshaped [n,n], tril."""
    x = np.asarray(x)
    m = x.shape[-1]
    n = int(np.sqrt(0.25 + 2.0 * m) - 0.5)
    assert n * (n + 1) // 2 == m
    xc = np.concatenate([x[..., n:], x[..., ::-1]], axis=-1)
    y = xc.reshape(x.shape[:-1] + (n, n))
    return np.tril(y)


def fill_triangular_inverse(y):
    """tfp.math.fill_triangular_inverse (lower): direct inversion of the forward index map."""
    y = np.asarray(y)
    n = y.shape[-1]
    m = n * (n + 1) // 2
    idx = np.arange(m)
    grid = np.concatenate([idx[n:], idx[::-1]]).reshape(n, n)  # (row, col) -> index in x
    out = np.zeros(y.shape[:-2] + (m,), dtype=y.dtype)
    r, c = np.tril_indices(n)
    out[..., grid[r, c]] = y[..., r, c]
    return out
