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
    """chol of a symmetric [M, M] (lower triangle read) with identity rows: returns (C lower, C^-T upper, solved extra)."""
    M = Sym.shape[0]
    e = 0 if extra is None else extra.shape[0]
    T = torch.empty((2 * M + e, M), dtype=torch.float64, device=Sym.device)
    T[:M] = Sym
    if e:
        T[M:M + e] = extra
    _, info = ops.potrf_(T, M, zero_upper=True, identity_rows=True)
    ops.check_info(info, "natural-gradient precision (step too long?): Cholesky")
    return T[:M], T[M + e:], (T[M:M + e] if e else None)


def natgrad_update(q_mu: torch.Tensor, q_sqrt: torch.Tensor, g_mu: torch.Tensor, g_sqrt: torch.Tensor, gamma: float
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """One step  theta <- theta - gamma dLoss/deta  for q_mu [M, P], q_sqrt [P, M, M]; g_* are gradients of the LOSS
    w.r.t. the (constrained) q_mu and q_sqrt.  Returns the new (q_mu, q_sqrt)."""
    M, P = q_mu.shape
    new_mu = torch.empty_like(q_mu)
    new_sqrt = torch.zeros_like(q_sqrt)
    for p in range(P):
        L = torch.tril(q_sqrt[p]).contiguous()
        mu_t = q_mu[:, p].reshape(1, M).contiguous()
        gmu_t = g_mu[:, p].reshape(1, M).contiguous()
        S = ops.gemm_nt(L, L, b_tri=2, a_tri=2)                                   # L L^T (L[j, kk] = 0 for kk > j)
        # Lc = chol(S): equals L when L's diagonal is positive.  The reference pushes g_L through
        # expectation_to_meanvarsqrt (natgrad.py:484-487, :327-329), i.e. through THIS factor, whatever the signs of
        # q_sqrt's diagonal -- mirrored here (a q_sqrt with negative diagonal entries gets the same step as there).
        Lc, LcinvT, _ = _factor_with_inverse(S)
        Sinv = ops.gemm_nt(LcinvT, LcinvT, b_tri=1, a_tri=1)                       # S^-1 = Lc^-T Lc^-1
        G = gradients.cholesky_adjoint(ops.transpose(Lc, mode=1), LcinvT, torch.tril(g_sqrt[p]).contiguous())
        Gmu_t = ops.gemm_nt(mu_t, G)                                      # (G mu)^T  (G symmetric)
        th1_t = ops.gemm_nt(mu_t, Sinv) - gamma * (gmu_t - 2.0 * Gmu_t)   # theta1'^T
        Pm = Sinv + (2.0 * gamma) * G                                     # -2 theta2'
        _, CinvT, w_t = _factor_with_inverse(Pm, th1_t)                   # C = chol(Pm); w^T = theta1'^T C^-T
        Snew = ops.gemm_nt(CinvT, CinvT, b_tri=1, a_tri=1)                         # S' = C^-T C^-1
        new_mu[:, p] = ops.gemm_nt(w_t.contiguous(), CinvT, b_tri=1).reshape(-1)   # mu' = C^-T (C^-1 theta1')
        Tn = Snew.clone()
        _, info = ops.potrf_(Tn, M, zero_upper=True)                      # L' = chol(S')
        ops.check_info(info, "natural-gradient covariance: Cholesky")
        new_sqrt[p] = Tn
    return new_mu, new_sqrt
