"""CPU oracle for the GRADIENTS of the SVGP hot path: a torch-CPU fp64 restatement of the same reference
formulas as oracle/gp_oracle.py, differentiated by torch autograd -- the stand-in for the TensorFlow autodiff
the reference uses (`optimizers/scipy.py:322-331`, `models/training_mixins.py:59-78`).

TEST INFRASTRUCTURE ONLY (same rules as gp_oracle.py: imported by tests/ only, never by gpflow_amd/).
Pinned by tests/test_oracle.py: the VALUE equals gp_oracle.svgp_elbo to 1e-12 (the NumPy restatement, itself pinned
to the reference's own source through tests/golden/ref_golden.npz: tests/test_reference_golden.py), and every gradient
agrees with central finite differences of gp_oracle.svgp_elbo.  The reference's gradients themselves (TensorFlow
autodiff) cannot be produced here; the pin is value-of-the-reference + derivative-of-that-value.
"""
from __future__ import annotations

import numpy as np
import torch

LOG2PI = float(np.log(2 * np.pi))


def _sqdist(X, X2):
    """gpflow/utilities/ops.py:105-122 (expansion formula)."""
    Xs = (X * X).sum(-1)
    X2s = (X2 * X2).sum(-1)
    return -2.0 * X @ X2.T + Xs[:, None] + X2s[None, :]


def _rbf(X, X2, variance, ls, family="SquaredExponential"):
    """stationaries.py:77-79, 103-116; K_r2 of SquaredExponential :209-210, K_r of Matern12 :254-255, Matern32 :281-283,
    Matern52 :311-313 (r = sqrt(max(r2, 1e-36)), :113-114)"""
    r2 = _sqdist(X / ls, X2 / ls)
    if family == "SquaredExponential":
        return variance * torch.exp(-0.5 * r2)
    r = torch.sqrt(torch.clamp(r2, min=1e-36))
    if family == "Matern12":
        return variance * torch.exp(-r)
    if family == "Matern32":
        s3 = float(np.sqrt(3.0))
        return variance * (1.0 + s3 * r) * torch.exp(-s3 * r)
    if family == "Matern52":
        s5 = float(np.sqrt(5.0))
        return variance * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * torch.exp(-s5 * r)
    raise KeyError(family)


def combination_kernel(members, op, cols=None):
    """(k(A, B), k_diag) of a flat Sum ("add") / Product ("mul") of stationary members [(family, variance, lengthscales)] on
    torch tensors: kernels/base.py:216-220 (tf.add_n of the member matrices), :305-315 (their elementwise product).  cols[i]: the
    input columns member i sees (its active_dims, kernels/base.py:90-109; each member slices for itself, :283-293), None = all."""
    cols = [None] * len(members) if cols is None else cols
    # op a string: flat; else a tree (op, [children]) over member indices -- nested Combinations (kernels/base.py:223-329)
    tree = (op, list(range(len(members)))) if isinstance(op, str) else op
    sl = lambda T_, c: T_ if c is None else T_[:, list(c)]  # noqa: E731

    def reduce(node, leaf):
        if isinstance(node, (int, np.integer)):
            return leaf(int(node))
        o, ch = node
        vals = [reduce(c, leaf) for c in ch]
        out = vals[0]
        for v_ in vals[1:]:
            out = out + v_ if o == "add" else out * v_
        return out

    def kfun(A, Bm):
        return reduce(tree, lambda i: _rbf(sl(A, cols[i]), sl(Bm, cols[i]), members[i][1], members[i][2], members[i][0]))
    kd = reduce(tree, lambda i: members[i][1])
    return kfun, kd


def _nvcol(noise_variance):
    """a constant, or one noise variance per row [B] -> [B, 1] (broadcasts against [B, P]; likelihoods/scalar_continuous.py:92-105)"""
    return noise_variance[:, None] if (torch.is_tensor(noise_variance) and noise_variance.dim() == 1) else noise_variance


def svgp_elbo_torch(X, Y, Z, q_mu, q_sqrt, variance, lengthscales, noise_variance, *, num_data=None, jitter=1e-6,
                    mean=0.0, whiten=True, family="SquaredExponential", kfun=None, kdiag=None):
    """SVGP.elbo (svgp.py:166-181) on torch fp64 tensors; q_sqrt [P, M, M]; whiten as in the reference.  kfun / kdiag: a
    kernel combination (combination_kernel) instead of the single stationary kernel."""
    M = Z.shape[0]
    B = X.shape[0]
    noise_variance = _nvcol(noise_variance)
    if kfun is None:
        kfun = lambda A_, B_: _rbf(A_, B_, variance, lengthscales, family)  # noqa: E731
    else:
        variance = kdiag
    Kmm = kfun(Z, Z) + jitter * torch.eye(M, dtype=torch.float64)                            # covariances/kuus.py:29-34
    Kmn = kfun(Z, X)                                                                         # kufs.py:31-34
    Lm = torch.linalg.cholesky(Kmm)                                                         # conditionals/util.py:67
    A = torch.linalg.solve_triangular(Lm, Kmn, upper=False)                                 # :125
    fvar = variance - (A * A).sum(0)                                                        # :133 (Knn = K_diag)
    if not whiten:
        A = torch.linalg.solve_triangular(Lm.T, A, upper=True)                              # :137-139
    fmean = A.T @ q_mu + mean                                                               # :144
    if q_sqrt.dim() == 2:   # q_diag: q_sqrt [M, P] standard deviations (:147-149, 164; kullback_leiblers.py:131-148)
        LTA = A[None, :, :] * q_sqrt.T[:, :, None]                                          # [P, M, B]
        fvar = (fvar[None, :] + (LTA * LTA).sum(1)).T
        ve = -0.5 * LOG2PI - 0.5 * torch.log(noise_variance) - 0.5 * ((Y - fmean) ** 2 + fvar) / noise_variance
        if whiten:
            kl = 0.5 * ((q_mu * q_mu).sum() - M * q_mu.shape[1] - torch.log(q_sqrt ** 2).sum() + (q_sqrt ** 2).sum())
        else:   # kullback_leiblers.py:107-165 with K = Kuu and a diagonal q: trace = sum diag(K^-1)[:, None] * q_sqrt^2 (:146-148)
            alpha = torch.linalg.solve_triangular(Lm, q_mu, upper=False)
            Lm_inv = torch.linalg.solve_triangular(Lm, torch.eye(M, dtype=torch.float64), upper=False)
            K_inv = torch.linalg.solve_triangular(Lm.T, Lm_inv, upper=True)
            kl = 0.5 * ((alpha * alpha).sum() - M * q_mu.shape[1] - torch.log(q_sqrt ** 2).sum()
                        + (torch.diagonal(K_inv)[:, None] * q_sqrt ** 2).sum()
                        + q_mu.shape[1] * torch.log(torch.diagonal(Lm) ** 2).sum())
        scale = 1.0 if num_data is None else float(num_data) / B
        return ve.sum() * scale - kl
    Lq = torch.tril(q_sqrt)                                                                 # :151
    LTA = Lq.transpose(1, 2) @ A                                                            # :157  [P, M, B]
    fvar = fvar[None, :] + (LTA * LTA).sum(1)                                               # :164  [P, B]
    fvar = fvar.T                                                                           # [B, P]
    ve = -0.5 * LOG2PI - 0.5 * torch.log(noise_variance) - 0.5 * ((Y - fmean) ** 2 + fvar) / noise_variance
    # gauss_kl, white (kullback_leiblers.py:98-165): 0.5 (mahalanobis - M P - sum log diag(Lq)^2 + trace)
    if whiten:
        kl = 0.5 * ((q_mu * q_mu).sum() - M * q_mu.shape[1]
                    - torch.log(torch.diagonal(Lq, dim1=1, dim2=2) ** 2).sum() + (Lq * Lq).sum())
    else:   # K = Kuu (kullback_leiblers.py:107-165, prior_kl :48-49)
        alpha = torch.linalg.solve_triangular(Lm, q_mu, upper=False)
        LpiLq = torch.linalg.solve_triangular(Lm.expand(Lq.shape[0], M, M), Lq, upper=False)
        kl = 0.5 * ((alpha * alpha).sum() - M * q_mu.shape[1] - torch.log(torch.diagonal(Lq, dim1=1, dim2=2) ** 2).sum()
                    + (LpiLq * LpiLq).sum() + q_mu.shape[1] * torch.log(torch.diagonal(Lm) ** 2).sum())
    scale = 1.0 if num_data is None else float(num_data) / B
    return ve.sum() * scale - kl


def svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, *, variance, lengthscales, noise_variance, num_data=None,
                              jitter=1e-6, mean=0.0, whiten=True, family="SquaredExponential"):
    """NumPy in, (value, dict of NumPy gradients w.r.t. the constrained quantities) out."""
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64, requires_grad=g)  # noqa: E731
    Zt, qm, qs = t(Z, True), t(q_mu, True), t(q_sqrt, True)
    var, ls, nv, mc = t(variance, True), t(np.atleast_1d(lengthscales), True), t(noise_variance, True), t(mean, True)
    F = svgp_elbo_torch(t(X), t(Y), Zt, qm, qs, var, ls, nv, num_data=num_data, jitter=jitter, mean=mc, whiten=whiten,
                        family=family)
    F.backward()
    g = {"variance": var.grad, "lengthscales": ls.grad, "noise_variance": nv.grad, "Z": Zt.grad, "q_mu": qm.grad,
         "q_sqrt": qs.grad, "mean_const": mc.grad}
    return float(F.detach()), {k: v.detach().numpy().copy() for k, v in g.items()}


def gpr_lml_torch(X, Y, variance, lengthscales, noise_variance, mean=0.0, family="SquaredExponential", kfun=None):
    """GPR.log_marginal_likelihood (gpr.py:91-107; logdensities.py:139-156) on torch fp64 tensors."""
    N = X.shape[0]
    Kxx = kfun(X, X) if kfun is not None else _rbf(X, X, variance, lengthscales, family)
    nvd = noise_variance if (torch.is_tensor(noise_variance) and noise_variance.dim() == 1) else noise_variance * torch.ones(N, dtype=torch.float64)
    K = Kxx + torch.diag(nvd)                                                                    # gpr.py:100-101, model_utils.py:46-50
    L = torch.linalg.cholesky(K)                                                                 # :102
    alpha = torch.linalg.solve_triangular(L, Y - mean, upper=False)                              # logdensities.py:150
    P = Y.shape[1]
    return -0.5 * (alpha * alpha).sum() - 0.5 * N * P * LOG2PI - P * torch.log(torch.diagonal(L)).sum()


def gpr_lml_value_and_grads(X, Y, *, variance, lengthscales, noise_variance, mean=0.0, family="SquaredExponential"):
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64, requires_grad=g)  # noqa: E731
    var, ls, nv, mc = t(variance, True), t(np.atleast_1d(lengthscales), True), t(noise_variance, True), t(mean, True)
    F = gpr_lml_torch(t(X), t(Y), var, ls, nv, mc, family)
    F.backward()
    g = {"variance": var.grad, "lengthscales": ls.grad, "noise_variance": nv.grad, "mean_const": mc.grad}
    return float(F.detach()), {k: v.detach().numpy().copy() for k, v in g.items()}


def combination_value_and_grads(model, X, Y, members, op, *, noise_variance, Z=None, q_mu=None, q_sqrt=None, num_data=None,
                                jitter=1e-6, mean=0.0, cols=None):
    """Value and gradients of GPR.log_marginal_likelihood ("gpr"), SVGP.elbo whitened ("svgp") / un-whitened ("svgp_unwhitened") or
    SGPR.elbo ("sgpr") under a Sum / Product of stationary kernels (members over the columns `cols`), by autograd:
    {"variance": [n], "lengthscales": [per member], "noise_variance", "Z", "q_mu", "q_sqrt"}."""
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64, requires_grad=g)  # noqa: E731
    vs = [t(v, True) for _, v, _ in members]
    lss = [t(np.atleast_1d(ls), True) for _, _, ls in members]
    kfun, kd = combination_kernel([(f, v, ls) for (f, _, _), v, ls in zip(members, vs, lss)], op, cols)
    nv = t(noise_variance, True)
    out = {}
    if model == "gpr":
        F = gpr_lml_torch(t(X), t(Y), None, None, nv, t(mean), kfun=kfun)
    elif model == "sgpr":
        Zt = t(Z, True)
        F = sgpr_elbo_torch(t(X), t(Y), Zt, None, None, nv, jitter=jitter, mean=t(mean), kfun=kfun, kdiag=kd)
    else:
        Zt, qm, qs = t(Z, True), t(q_mu, True), t(q_sqrt, True)
        F = svgp_elbo_torch(t(X), t(Y), Zt, qm, qs, None, None, nv, num_data=num_data, jitter=jitter, mean=t(mean),
                            whiten=(model == "svgp"), kfun=kfun, kdiag=kd)
    F.backward()
    out.update(variance=np.array([float(v.grad) for v in vs]), lengthscales=[ls.grad.numpy().copy() for ls in lss],
               noise_variance=float(nv.grad))
    if model != "gpr":
        out.update(Z=Zt.grad.numpy().copy())
    if model in ("svgp", "svgp_unwhitened"):
        out.update(q_mu=qm.grad.numpy().copy(), q_sqrt=qs.grad.numpy().copy())
    return float(F.detach()), out


# ----------------------------------------------------------------------------- natural gradient
def natural_to_meanvarsqrt_torch(nat1, nat2):
    """gpflow/optimizers/natgrad.py:429-441 on [P, M, 1] / [P, M, M] torch tensors"""
    vsi = torch.linalg.cholesky(-2 * nat2)
    vs = torch.linalg.solve_triangular(vsi, torch.eye(vsi.shape[1], dtype=torch.float64).expand_as(vsi), upper=False)
    S = vs.transpose(1, 2) @ vs
    return S @ nat1, torch.linalg.cholesky(S)


def natgrad_conversions(q_mu, q_sqrt):
    """The parameter conversions of gpflow/optimizers/natgrad.py:429-516 on NumPy q_mu [M, P], q_sqrt [P, M, M] (in the
    reference's [N, D] / [D, N, N] layout): returns a dict with nat1, nat2 (meanvarsqrt_to_natural :444-455), eta1, eta2
    (meanvarsqrt_to_expectation :490-498) and the round trips natural_to_meanvarsqrt (:429-441) /
    expectation_to_meanvarsqrt (:479-487).  Pinned to the reference's own functions (tests/test_reference_golden.py)."""
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64)  # noqa: E731
    mu = t(q_mu).T[:, :, None]
    Ls = torch.tril(t(q_sqrt))
    Linv = torch.linalg.solve_triangular(Ls, torch.eye(Ls.shape[1], dtype=torch.float64).expand_as(Ls), upper=False)
    s_inv = Linv.transpose(1, 2) @ Linv
    nat1, nat2 = s_inv @ mu, -0.5 * s_inv
    eta1, eta2 = mu, Ls @ Ls.transpose(1, 2) + mu @ mu.transpose(1, 2)
    m_b, s_b = natural_to_meanvarsqrt_torch(nat1, nat2)
    s_c = torch.linalg.cholesky(eta2 - eta1 @ eta1.transpose(1, 2))
    back = lambda v: v[:, :, 0].T.numpy().copy()  # noqa: E731
    return {"nat1": back(nat1), "nat2": nat2.numpy().copy(), "eta1": back(eta1), "eta2": eta2.numpy().copy(),
            "back_mu": back(m_b), "back_sqrt": s_b.numpy().copy(), "back2_mu": back(eta1), "back2_sqrt": s_c.numpy().copy()}


def natgrad_step(q_mu, q_sqrt, g_mu, g_sqrt, gamma, xi_transform="XiNat"):
    """One XiNat step of gpflow/optimizers/natgrad.py:280-368, restated literally with torch autograd standing in for
    the TF tapes: dL/deta through expectation_to_meanvarsqrt (:484-487), theta <- theta - gamma dL/deta (:341-343),
    natural_to_meanvarsqrt (:429-441).  q_mu [M, P], q_sqrt [P, M, M]; g_* = loss gradients w.r.t. them."""
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64)  # noqa: E731
    mu = t(q_mu).T[:, :, None]                               # [P, M, 1]   (swap_dimensions :385-420)
    Ls = torch.tril(t(q_sqrt))
    gm = t(g_mu).T[:, :, None]
    gL = torch.tril(t(g_sqrt))
    eta1 = mu.clone().requires_grad_(True)
    eta2 = (Ls @ Ls.transpose(1, 2) + mu @ mu.transpose(1, 2)).requires_grad_(True)       # :496-498
    var = eta2 - eta1 @ eta1.transpose(1, 2)                                             # :485
    m_out, s_out = eta1, torch.linalg.cholesky(var)                                      # :486-487
    dL_deta1, dL_deta2 = torch.autograd.grad([m_out, s_out], [eta1, eta2], grad_outputs=[gm, gL])   # :327-329
    Linv = torch.linalg.solve_triangular(Ls, torch.eye(Ls.shape[1], dtype=torch.float64).expand_as(Ls), upper=False)
    s_inv = Linv.transpose(1, 2) @ Linv                                                  # :452-454
    nat1, nat2 = s_inv @ mu, -0.5 * s_inv
    if xi_transform == "XiSqrtMeanVar":
        # xi = natural_to_meanvarsqrt(theta) (:139-173): nat_dL_xi = (d xi / d theta) dL/deta, a forward-mode product (:323-339)
        (_, _), (dxi1, dxi2) = torch.func.jvp(natural_to_meanvarsqrt_torch, (nat1.detach(), nat2.detach()),
                                              (dL_deta1.detach(), dL_deta2.detach()))
        mun, Ln = mu - gamma * dxi1, Ls - gamma * dxi2                                   # :341-347 with xi = (mean, varsqrt)
        return mun[:, :, 0].T.numpy().copy(), Ln.numpy().copy()
    nat1n, nat2n = nat1 - gamma * dL_deta1, nat2 - gamma * dL_deta2                      # :341-343
    vsi = torch.linalg.cholesky(-2 * nat2n)                                              # :435
    vs = torch.linalg.solve_triangular(vsi, torch.eye(vsi.shape[1], dtype=torch.float64).expand_as(vsi), upper=False)
    S = vs.transpose(1, 2) @ vs                                                          # :437
    mun = S @ nat1n                                                                      # :438
    return mun[:, :, 0].T.numpy().copy(), torch.linalg.cholesky(S).numpy().copy()        # :441


# ----------------------------------------------------------------------------- SGPR gradients (SURVEY 8f rows 1 + 3)
def sgpr_elbo_torch(X, Y, Z, variance, lengthscales, noise_variance, *, jitter=1e-6, mean=0.0,
                    family="SquaredExponential", kfun=None, kdiag=None):
    """SGPR.elbo (gpflow/models/sgpr.py:181-290) on torch fp64 tensors; noise_variance a scalar or one value per data row [N]
    (likelihood.variance_at(X), :207).  kfun / kdiag: a kernel combination (combination_kernel) instead of the single stationary kernel."""
    N, P = Y.shape
    M = Z.shape[0]
    nvr = noise_variance.reshape(-1).expand(N) if noise_variance.numel() in (1, N) else noise_variance
    sigma = torch.sqrt(nvr)
    if kfun is None:
        kfun = lambda A_, B_: _rbf(A_, B_, variance, lengthscales, family)  # noqa: E731
    else:
        variance = kdiag
    kuf = kfun(Z, X)
    kuu = kfun(Z, Z) + jitter * torch.eye(M, dtype=torch.float64)
    L = torch.linalg.cholesky(kuu)
    A = torch.linalg.solve_triangular(L, kuf / sigma, upper=False)
    AAT = A @ A.T
    LB = torch.linalg.cholesky(AAT + torch.eye(M, dtype=torch.float64))
    trace = (variance / nvr).sum() - torch.trace(AAT)                                            # :236-242
    logdet = -P * (torch.log(torch.diagonal(LB)).sum() + 0.5 * torch.log(nvr).sum() + 0.5 * trace)   # :245-251
    err = (Y - mean) / sigma[:, None]
    c = torch.linalg.solve_triangular(LB, A @ err, upper=False)
    quad = -0.5 * ((err * err).sum() - (c * c).sum())
    return -0.5 * N * P * LOG2PI + logdet + quad


def sgpr_elbo_value_and_grads(X, Y, Z, *, variance, lengthscales, noise_variance, jitter=1e-6, mean=0.0,
                              family="SquaredExponential"):
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64, requires_grad=g)  # noqa: E731
    Zt = t(Z, True)
    var, ls, nv, mc = t(variance, True), t(np.atleast_1d(lengthscales), True), t(noise_variance, True), t(mean, True)
    F = sgpr_elbo_torch(t(X), t(Y), Zt, var, ls, nv, jitter=jitter, mean=mc, family=family)
    F.backward()
    g = {"variance": var.grad, "lengthscales": ls.grad, "noise_variance": nv.grad, "Z": Zt.grad, "mean_const": mc.grad}
    return float(F.detach()), {k: v.detach().numpy().copy() for k, v in g.items()}


def heteroskedastic_value_and_grads(model, X, Y, *, A, b, variance=None, lengthscales=None, Z=None, q_mu=None, q_sqrt=None, num_data=None,
                                    jitter=1e-6, mean=0.0, lower_bound=1e-6, combination=None):
    """GPR.log_marginal_likelihood ("gpr"), SGPR.elbo ("sgpr") or SVGP.elbo whitened ("svgp") / un-whitened ("svgp_unwhitened") under Gaussian(scale=Linear(A, b))
    (likelihoods/scalar_continuous.py:52-111: sigma_n^2 = max(x_n A + b, sqrt(lower bound))^2) and their gradients w.r.t. A, b and
    the other parameters, by autograd."""
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64, requires_grad=g)  # noqa: E731
    At, bt = t(np.atleast_2d(A), True), t(np.atleast_1d(b), True)
    Xt = t(X)
    nv = torch.clamp(Xt @ At + bt, min=float(np.sqrt(lower_bound)))[:, 0] ** 2
    out = {}
    kw = {}
    if combination is not None:     # (members, op or tree, cols): a Sum / Product of stationary kernels under the same likelihood
        members, op, cols = combination
        vs = [t(v, True) for _, v, _ in members]
        lss = [t(np.atleast_1d(l_), True) for _, _, l_ in members]
        kfun, kd = combination_kernel([(f, v, l_) for (f, _, _), v, l_ in zip(members, vs, lss)], op, cols)
        var = ls = None
        kw = dict(kfun=kfun, kdiag=kd) if model != "gpr" else dict(kfun=kfun)
    else:
        var, ls = t(variance, True), t(np.atleast_1d(lengthscales), True)
    if model == "gpr":
        F = gpr_lml_torch(Xt, t(Y), var, ls, nv, t(mean), **kw)
    elif model == "sgpr":
        Zt = t(Z, True)
        F = sgpr_elbo_torch(Xt, t(Y), Zt, var, ls, nv, jitter=jitter, mean=t(mean), **kw)
    else:
        Zt, qm, qs = t(Z, True), t(q_mu, True), t(q_sqrt, True)
        F = svgp_elbo_torch(Xt, t(Y), Zt, qm, qs, var, ls, nv, num_data=num_data, jitter=jitter, mean=t(mean), whiten=(model == "svgp"), **kw)
    F.backward()
    out.update(A=At.grad.numpy().copy(), b=bt.grad.numpy().copy())
    if combination is not None:
        out.update(variance=np.array([float(v.grad) for v in vs]), lengthscales=[l_.grad.numpy().copy() for l_ in lss])
    else:
        out.update(variance=float(var.grad), lengthscales=ls.grad.numpy().copy())
    if model == "sgpr":
        out.update(Z=Zt.grad.numpy().copy())
    elif model != "gpr":
        out.update(Z=Zt.grad.numpy().copy(), q_mu=qm.grad.numpy().copy(), q_sqrt=qs.grad.numpy().copy())
    return float(F.detach()), out
