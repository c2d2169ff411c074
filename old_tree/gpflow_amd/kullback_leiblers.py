"""KL[q || p] for the variational posterior (gpflow/kullback_leiblers.py:28-165)."""
from __future__ import annotations

from typing import Optional

import torch

from . import config, ops
from .covariances import Kuu


def prior_kl(inducing_variable, kernel, q_mu, q_sqrt, whiten: bool = False) -> torch.Tensor:
    """kullback_leiblers.py:31-49"""
    if whiten:
        return gauss_kl(q_mu, q_sqrt, None)
    K = Kuu(inducing_variable, kernel, jitter=config.default_jitter())
    return gauss_kl(q_mu, q_sqrt, K)


def gauss_kl(q_mu, q_sqrt, K=None, *, K_cholesky=None) -> torch.Tensor:
    """kullback_leiblers.py:59-165.  q_mu [M,L]; q_sqrt [L,M,M] (lower; upper part ignored) or [M,L];
    K / K_cholesky [M,M] or [L,M,M] or None (white).  Returns a 0-dim device tensor."""
    if (K is not None) and (K_cholesky is not None):
        raise ValueError(
            "Ambiguous arguments: gauss_kl() must only be passed one of `K` or `K_cholesky`.")
    q_mu = ops.to_device(q_mu)
    q_sqrt = ops.to_device(q_sqrt)
    is_white = (K is None) and (K_cholesky is None)
    is_diag = q_sqrt.dim() == 2
    M, L = q_mu.shape
    if is_white:
        return ops.gauss_kl_white(q_mu, q_sqrt)[0]

    Kin = ops.to_device(K if K is not None else K_cholesky)
    is_batched = Kin.dim() == 3
    # right-hand sides that ride through the factorisation as extra rows:
    #   q_mu[:, l]^T                      -> alpha^T            (kullback_leiblers.py:114)
    #   tril(q_sqrt_l)^T  (M rows)        -> (Lp^-1 Lq_l)^T     (:152)   [full q_sqrt]
    #   I                 (M rows)        -> Lp^-T              (:139-143) [q_diag, shared K]
    if is_diag:
        Lq_diag = q_sqrt
        if is_batched:
            rhs_blocks = [torch.diag_embed(q_sqrt.t().contiguous())]  # [L,M,M] diag matrices
        else:
            rhs_blocks = None
    else:
        LqT = ops.transpose(q_sqrt, mode=1)  # [L,M,M], = tril(q_sqrt_l)^T
        Lq_diag = torch.diagonal(q_sqrt, dim1=-2, dim2=-1).t()  # [M,L]
        rhs_blocks = [LqT]
    dev = q_mu.device
    if is_batched:
        n_extra = 1 + M
        T = torch.empty((L, M + n_extra, M), dtype=torch.float64, device=dev)
        T[:, M] = q_mu.t()
        T[:, M + 1:] = rhs_blocks[0]
    else:
        if is_diag:
            n_extra = L + M
            T = torch.empty((M + n_extra, M), dtype=torch.float64, device=dev)
            T[M:M + L] = q_mu.t()
            if K is None:   # (with K the factorisation writes the identity rows itself: ops.potrf_(identity_rows=True))
                T[M + L:] = torch.eye(M, dtype=torch.float64, device=dev)
        else:
            n_extra = L + L * M
            T = torch.empty((M + n_extra, M), dtype=torch.float64, device=dev)
            T[M:M + L] = q_mu.t()
            T[M + L:] = rhs_blocks[0].reshape(L * M, M)
    if K is not None:
        if is_batched:
            T[:, :M] = Kin
        else:
            T[:M] = Kin
        _, info = ops.potrf_(T, M, identity_rows=(is_diag and not is_batched))
        ops.check_info(info)
        Lp = T[:, :M] if is_batched else T[:M]
    else:
        Lp = Kin
        if is_batched:
            for l in range(L):
                invd = ops.trtri_blocks(Lp[l].contiguous())
                ops.trsm_(T[l, M:], Lp[l].contiguous(), invd, trans=0)
        else:
            invd = ops.trtri_blocks(Lp.contiguous())
            ops.trsm_(T[M:], Lp.contiguous(), invd, trans=0)
    if is_batched:
        flat = T[:, M:].reshape(L * n_extra, M) if T[:, M:].is_contiguous() else T[:, M:].contiguous().reshape(L * n_extra, M)
        alpha_rows = T[:, M].contiguous()
        mahalanobis = ops.sumsq(alpha_rows)[0]
        trace = ops.sumsq(flat)[0] - mahalanobis
        sum_log_diag_Lp = ops.sum_log_diag(Lp).sum()
        scale = 1.0
    else:
        mahalanobis = ops.sumsq(T[M:M + L])[0]
        if is_diag:
            kinv_diag, _, _ = ops.row_stats(T[M + L:])  # diag(K^-1)_k = sum_i (Lp^-1)[i,k]^2
            trace = (kinv_diag[:, None] * q_sqrt ** 2).sum()
        else:
            trace = ops.sumsq(T[M + L:])[0]
        sum_log_diag_Lp = ops.sum_log_diag(Lp)[0]
        scale = float(L)
    constant = -float(M * L)
    logdet_qcov = torch.log(Lq_diag ** 2).sum()
    twoKL = mahalanobis + constant - logdet_qcov + trace
    twoKL = twoKL + scale * 2.0 * sum_log_diag_Lp  # sum log(diag(Lp)^2) (kullback_leiblers.py:159-163)
    return 0.5 * twoKL
