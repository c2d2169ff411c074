"""CPU emulation of the `gpflow_amd.ops` primitives -- TEST INFRASTRUCTURE ONLY.

`gpflow_amd/gradients.py` is a composition of device primitives.  There is no GPU in the build container, so the
composition itself (operand order, transposes, which triangle / K range each GEMM may assume) is validated here by
swapping `gradients.ops` for this module in a CPU test and comparing with the autograd oracle.  The emulation
reproduces the CONTRACT of each primitive, including what the HIP kernels do NOT compute:
  * gemm_nt(b_tri=1): for column tile n0 the K range starts at n0 & ~15 -- B[j, kk < n0] is never read;
    b_tri=2: the K range ends at n0 + 128 -- B[j, kk >= n0 + 128] is never read (poisoned with NaN here);
  * gemm_nt(c_lower): 128 x 128 tiles strictly above the diagonal are skipped (left at their initial value);
  * potrf_ reads only the lower triangle of the square block and leaves / zeroes the upper one;
  * transpose(mode=1) keeps only the lower triangle of its input.
The product never imports this file.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import torch

NB = 128


def _np(t):
    return t.detach().cpu().numpy()


def to_device(x, dtype=torch.float64):
    if isinstance(x, torch.Tensor):
        return x.to(dtype=dtype).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=dtype).contiguous()


def _ls(lengthscales, d):
    return np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (d,))


def _k(X1, X2, variance, lengthscales):
    a, b = _np(X1) / _ls(lengthscales, X1.shape[1]), _np(X2) / _ls(lengthscales, X1.shape[1])
    r2 = -2.0 * a @ b.T + (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :]
    return variance * np.exp(-0.5 * r2)


def kernel_matrix(X1, X2, *, variance, lengthscales, family="SquaredExponential", diag_add=0.0, lower_only=False,
                  out=None):
    assert family == "SquaredExponential"
    K = _k(X1, X1 if X2 is None else X2, variance, lengthscales)
    if X2 is None:
        K = K + diag_add * np.eye(K.shape[0])
        if lower_only:   # tiles strictly above the diagonal are not written
            K = np.where(np.triu(np.ones_like(K), 1) > 0, np.nan, K)
    Kt = torch.from_numpy(K)
    if out is None:
        return Kt
    out.copy_(Kt)
    return out


def kernel_matrix_hadamard(X1, X2, G, *, variance, lengthscales, family="SquaredExponential", out=None):
    assert family == "SquaredExponential" and tuple(G.shape) == (X1.shape[0], X2.shape[0])
    R = torch.from_numpy(_k(X1, X2, variance, lengthscales) * _np(G))
    if out is None:
        return R
    out.copy_(R)
    return out


def potrf_(T, n, *, zero_upper=False, invd=None):
    assert T.dim() == 2 and T.shape[1] == n and T.stride(1) == 1
    K = np.tril(_np(T[:n]))
    K = K + np.tril(K, -1).T          # only the lower triangle is read
    L = np.linalg.cholesky(K)
    E = _np(T[n:])
    S = sla.solve_triangular(L, E.T, lower=True).T if E.shape[0] else E
    up = _np(T[:n]) * np.triu(np.ones((n, n)), 1)
    T[:n] = torch.from_numpy(L + (0.0 if zero_upper else up))
    T[n:] = torch.from_numpy(S)
    return ("invd", n), torch.zeros(1, dtype=torch.int32)


def check_info(info, what="Cholesky"):
    assert int(info[0]) == 0


def transpose_factor(L, invd):
    assert invd == ("invd", L.shape[0])
    return torch.tril(L).t().contiguous(), ("invdT", L.shape[0])


def trsm_(B, L, invd, *, trans=0):
    """trans=0: B <- B L^-T given (L, invd); trans=1: B <- B L^-1 given (LT, invdT)."""
    n = L.shape[0]
    if trans == 0:
        assert invd == ("invd", n)
        Ll = np.tril(_np(L))
        B.copy_(torch.from_numpy(sla.solve_triangular(Ll, _np(B).T, lower=True).T))
    else:
        assert invd == ("invdT", n)
        Ll = np.triu(_np(L)).T      # the argument is L^T (upper); only that triangle is read
        # B L^-1 = (L^-T B^T)^T
        B.copy_(torch.from_numpy(sla.solve_triangular(Ll.T, _np(B).T, lower=False).T))
    return B


def gemm_nt(A, B, *, alpha=1.0, beta=0.0, C=None, b_tri=0, c_lower=False):
    batched = A.dim() == 3 or B.dim() == 3
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    B3 = B if B.dim() == 3 else B.unsqueeze(0)
    batch = max(A3.shape[0], B3.shape[0])
    m, k = A3.shape[1], A3.shape[2]
    n = B3.shape[1]
    assert B3.shape[2] == k
    if C is None:
        assert beta == 0.0
        C = torch.zeros((batch, m, n) if batched else (m, n), dtype=torch.float64)
        if not c_lower:
            C.fill_(float("nan"))     # every entry must be written by the kernel
    C3 = C if C.dim() == 3 else C.unsqueeze(0)
    for z in range(batch):
        a = _np(A3[z if A3.shape[0] > 1 else 0])
        b = _np(B3[z if B3.shape[0] > 1 else 0]).copy()
        for n0 in range(0, n, NB):
            n1 = min(n0 + NB, n)
            kb, ke = 0, k
            if b_tri == 1:
                kb = min(n0 & ~15, k)
            elif b_tri == 2:
                ke = min(n0 + NB, k)
            prod = a[:, kb:ke] @ b[n0:n1, kb:ke].T
            for m0 in range(0, m, NB):
                m1 = min(m0 + NB, m)
                if c_lower and n0 > m0 + NB - 1:
                    continue
                old = _np(C3[z, m0:m1, n0:n1])
                new = alpha * prod[m0:m1] + (beta * old if beta != 0.0 else 0.0)
                C3[z, m0:m1, n0:n1] = torch.from_numpy(new)
    return C


def transpose(X, *, mode=0, out=None):
    Y = X
    if mode == 1:
        Y = torch.tril(X)
    elif mode == 2:
        Y = torch.triu(X)
    R = Y.transpose(-1, -2).contiguous()
    if out is None:
        return R
    out.copy_(R)
    return out


def row_stats(At, *, V=None, W=None, want_sumsq=True):
    a = _np(At)
    sumsq = torch.from_numpy((a * a).sum(1)) if want_sumsq else None
    mv = torch.from_numpy(a @ _np(V)) if V is not None else None
    wsq = torch.from_numpy(((a * a) @ (_np(W) ** 2)).T.copy()) if W is not None else None
    return sumsq, mv, wsq


def gaussian_varexp_sum(Y, fmean, *, s0, ssq, knn, noise_variance, mean_const=0.0, s0_per_latent=False,
                        want_fvar=False):
    assert not s0_per_latent and len(knn) == 1
    fv = knn[0] - _np(s0)[:, None] + _np(ssq).T
    ve = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(noise_variance) \
        - 0.5 * ((_np(Y) - _np(fmean) - mean_const) ** 2 + fv) / noise_variance
    return torch.tensor([ve.sum()], dtype=torch.float64), (torch.from_numpy(fv) if want_fvar else None)


def gauss_kl_white(q_mu, q_sqrt):
    Lq = np.tril(_np(q_sqrt))
    M, P = q_mu.shape
    kl = 0.5 * ((_np(q_mu) ** 2).sum() - M * P - np.log(np.diagonal(Lq, axis1=1, axis2=2) ** 2).sum() + (Lq * Lq).sum())
    return torch.tensor([kl], dtype=torch.float64)


def sumsq(A, *, upper_only=False):
    a = _np(A)
    if upper_only:
        a = np.triu(a)
    return torch.tensor([(a * a).sum()], dtype=torch.float64)


def sum_log_diag(L):
    L3 = L if L.dim() == 3 else L.unsqueeze(0)
    return torch.from_numpy(np.log(np.diagonal(_np(L3), axis1=1, axis2=2)).sum(1))
