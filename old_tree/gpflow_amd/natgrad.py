"""Natural-gradient step for the Gaussian q(u) = N(q_mu, q_sqrt q_sqrt^T) of an SVGP (gpflow/optimizers/natgrad.py:280-368,
conversions :429-516), natural parametrisation (XiNat) -- SURVEY 8f row 3: "batched [P,M,M] cholesky, lower-triangular
inverse, matmuls -- again only potrf / trsm / gemm".

The reference differentiates its parameter conversions with TF; here the chain rule is written out.  Per latent, with
S = L L^T (L = tril(q_sqrt)), expectation parameters eta = (mu, S + mu mu^T), natural parameters
theta = (S^-1 mu, -S^-1 / 2) and loss gradients (g_mu, g_L) w.r.t. (mu, L):

    G        = sym(L^-T Phi(L^T g_L) L^-1)              dLoss/dS  (Cholesky adjoint, gradients.cholesky_adjoint; L = chol(S))
    dLoss/deta = (g_mu - 2 G mu,  G)
    theta1'  = S^-1 mu - gamma (g_mu - 2 G mu),         -2 theta2' = S^-1 + 2 gamma G
    S'       = (-2 theta2')^-1,   mu' = S' theta1',     L' = chol(S')              (natgrad.py:341-349, :429-441)

Every inverse comes out of the trapezoidal Cholesky (identity rows appended -> the factor's inverse transpose), so
the step is three factorisations of size M and a handful of M^3 triangular-K GEMMs per latent.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import gradients, ops


def _factor_with_inverse(Sym: torch.Tensor, extra: torch.Tensor = None):
    """Batched chol of symmetric [P, M, M] (lower triangles read) with identity rows appended: ONE gpk_potrf(batch = P)
    returns (C lower [P,M,M], C^-T upper [P,M,M], solved extra rows [P,e,M] or None) -- the batched [P, M, M] Cholesky /
    triangular inverse of gpflow/optimizers/natgrad.py:429-516."""
    P, M, _ = Sym.shape
    e = 0 if extra is None else extra.shape[1]
    T = torch.zeros((P, 2 * M + e, M), dtype=torch.float64, device=Sym.device)
    T[:, :M] = Sym
    if e:
        T[:, M:M + e] = extra
    idx = torch.arange(M, device=Sym.device)
    T[:, M + e + idx, idx] = 1.0
    _, info = ops.potrf_(T, M, zero_upper=True)
    ops.check_info(info, "natural-gradient precision (step too long?): Cholesky")
    return T[:, :M], T[:, M + e:], (T[:, M:M + e] if e else None)


def _phi(T: torch.Tensor) -> torch.Tensor:
    """Phi(T) for [P, M, M]: lower triangle with the diagonal halved."""
    out = torch.tril(T)
    d = torch.arange(T.shape[-1], device=T.device)
    out[:, d, d] *= 0.5
    return out


def _cholesky_adjoint(LT: torch.Tensor, LinvT: torch.Tensor, Lbar: torch.Tensor) -> torch.Tensor:
    """gradients.cholesky_adjoint for [P, M, M] stacks: sym(L^-T Phi(L^T L_bar) L^-1), three batched triangular-K GEMMs."""
    T1 = ops.gemm_nt(LT, Lbar.transpose(1, 2).contiguous(), b_tri=1, a_tri=1)          # L^T L_bar
    Y = ops.gemm_nt(_phi(T1), LinvT, b_tri=1, a_tri=2)                                  # Phi L^-1 (lower)
    S = ops.gemm_nt(LinvT, torch.tril(Y).transpose(1, 2).contiguous(), b_tri=1, a_tri=1)  # L^-T (Phi L^-1)
    return 0.5 * (S + S.transpose(1, 2))


XI_TRANSFORMS = ("XiNat", "XiSqrtMeanVar")


def natgrad_update(q_mu: torch.Tensor, q_sqrt: torch.Tensor, g_mu: torch.Tensor, g_sqrt: torch.Tensor, gamma: float,
                   xi_transform: str = "XiNat") -> Tuple[torch.Tensor, torch.Tensor]:
    """One natural-gradient step for q_mu [M, P], q_sqrt [P, M, M]; g_* are gradients of the LOSS w.r.t. the (constrained)
    q_mu and q_sqrt.  Returns the new (q_mu, q_sqrt).  All P latents in one batched launch sequence (no host loop).

    xi_transform (gpflow/optimizers/natgrad.py:98-173):
      "XiNat"          xi = natural parameters: theta <- theta - gamma dLoss/deta, then natural_to_meanvarsqrt (:341-349);
      "XiSqrtMeanVar"  xi = (q_mu, q_sqrt) itself: xi <- xi - gamma (d xi / d theta) dLoss/deta (:323-339, forward-mode through
                       natural_to_meanvarsqrt).  Written out, with S = L L^T, d1 = g_mu - 2 G mu, d2 = G:
                       d mu = S d1 + 2 S d2 mu = S g_mu,   d L = L Phi(L^-1 (2 S d2 S) L^-T) = L Phi(L^T g_L)
                       -- no factorisation at all: four batched triangular products.
    """
    if xi_transform not in XI_TRANSFORMS:
        raise NotImplementedError(f"xi_transform {xi_transform!r}: only {XI_TRANSFORMS} (natgrad.py:98-173)")
    M, P = q_mu.shape
    L = torch.tril(q_sqrt).contiguous()                                   # [P, M, M]
    gL = torch.tril(g_sqrt).contiguous()
    mu_t = q_mu.t().contiguous().reshape(P, 1, M)
    gmu_t = g_mu.t().contiguous().reshape(P, 1, M)
    LT = L.transpose(1, 2).contiguous()                                   # upper
    if xi_transform == "XiSqrtMeanVar":
        A = ops.gemm_nt(LT, gL.transpose(1, 2).contiguous(), b_tri=1, a_tri=1)          # L^T g_L
        Ph = _phi(A)
        dL = ops.gemm_nt(L, Ph.transpose(1, 2).contiguous(), b_tri=1, a_tri=2)          # L Phi(L^T g_L)
        t = ops.gemm_nt(gmu_t, LT)                                                      # g_mu^T L
        dmu_t = ops.gemm_nt(t, L)                                                       # (L L^T g_mu)^T
        new_mu = (mu_t - gamma * dmu_t).reshape(P, M).t().contiguous()
        return new_mu, torch.tril(L - gamma * dL)
    S = ops.gemm_nt(L, L, b_tri=2, a_tri=2)                                            # L L^T (L[j, kk] = 0 for kk > j)
    # Lc = chol(S): equals L when L's diagonal is positive.  The reference pushes g_L through
    # expectation_to_meanvarsqrt (natgrad.py:484-487, :327-329), i.e. through THIS factor, whatever the signs of
    # q_sqrt's diagonal -- mirrored here (a q_sqrt with negative diagonal entries gets the same step as there).
    Lc, LcinvT, _ = _factor_with_inverse(S)
    LcinvT = torch.triu(LcinvT).contiguous()
    Sinv = ops.gemm_nt(LcinvT, LcinvT, b_tri=1, a_tri=1)                                # S^-1 = Lc^-T Lc^-1
    G = _cholesky_adjoint(torch.tril(Lc).transpose(1, 2).contiguous(), LcinvT, gL)
    Gmu_t = ops.gemm_nt(mu_t, G)                                                        # (G mu)^T  (G symmetric)
    th1_t = ops.gemm_nt(mu_t, Sinv) - gamma * (gmu_t - 2.0 * Gmu_t)                     # theta1'^T
    Pm = Sinv + (2.0 * gamma) * G                                                       # -2 theta2'
    _, CinvT, w_t = _factor_with_inverse(Pm, th1_t)                                     # C = chol(Pm); w^T = theta1'^T C^-T
    CinvT = torch.triu(CinvT).contiguous()
    Snew = ops.gemm_nt(CinvT, CinvT, b_tri=1, a_tri=1)                                  # S' = C^-T C^-1
    new_mu = ops.gemm_nt(w_t.contiguous(), CinvT, b_tri=1).reshape(P, M).t().contiguous()   # mu' = C^-T (C^-1 theta1')
    Tn = Snew.clone()
    _, info = ops.potrf_(Tn, M, zero_upper=True)                                        # L' = chol(S'), batched
    ops.check_info(info, "natural-gradient covariance: Cholesky")
    return new_mu, torch.tril(Tn)
