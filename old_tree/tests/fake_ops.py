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
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)


KERNEL_FAMILIES = {"SquaredExponential": 0, "Matern12": 1, "Matern32": 2, "Matern52": 3}


def device():
    return torch.device("cpu")


def to_device(x, dtype=torch.float64):
    # always a private copy, like a host -> device transfer (in-place primitives must never reach the caller's array)
    if isinstance(x, torch.Tensor):
        return x.to(dtype=dtype).contiguous()
    return torch.tensor(np.array(x, dtype=np.float64, copy=True), dtype=dtype).contiguous()


def _ls(lengthscales, d):
    return np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (d,))


def _k(X1, X2, variance, lengthscales, family="SquaredExponential"):
    a, b = _np(X1) / _ls(lengthscales, X1.shape[1]), _np(X2) / _ls(lengthscales, X1.shape[1])
    r2 = -2.0 * a @ b.T + (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :]
    if family == "SquaredExponential":
        return variance * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, 1e-36))           # stationaries.py:113-114
    if family == "Matern12":
        return variance * np.exp(-r)
    if family == "Matern32":
        return variance * (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
    if family == "Matern52":
        return variance * (1.0 + np.sqrt(5.0) * r + 5.0 / 3.0 * r * r) * np.exp(-np.sqrt(5.0) * r)
    raise KeyError(family)


def kernel_matrix(X1, X2, *, variance, lengthscales, family="SquaredExponential", diag_add=0.0, lower_only=False,
                  out=None):
    K = _k(X1, X1 if X2 is None else X2, variance, lengthscales, family)
    if X2 is None:
        K = K + diag_add * np.eye(K.shape[0])
        if lower_only:   # tiles strictly above the diagonal are not written
            K = np.where(np.triu(np.ones_like(K), 1) > 0, np.nan, K)
    Kt = torch.from_numpy(K)
    if out is None:
        return Kt
    out.copy_(Kt)
    return out


def _dr2(X1, X2, variance, lengthscales, family):
    """-2 dk/dr2 at the scaled squared distance (zero where the 1e-36 clamp is active)."""
    a, b = _np(X1) / _ls(lengthscales, X1.shape[1]), _np(X2) / _ls(lengthscales, X1.shape[1])
    r2 = -2.0 * a @ b.T + (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :]
    if family == "SquaredExponential":
        return variance * np.exp(-0.5 * r2)
    ok = r2 > 1e-36
    r = np.sqrt(np.where(ok, r2, 1.0))
    if family == "Matern12":
        f = variance * np.exp(-r) / r
    elif family == "Matern32":
        f = 3.0 * variance * np.exp(-np.sqrt(3.0) * r)
    elif family == "Matern52":
        f = (5.0 / 3.0) * variance * (1.0 + np.sqrt(5.0) * r) * np.exp(-np.sqrt(5.0) * r)
    else:
        raise KeyError(family)
    return np.where(ok, f, 0.0)


def kernel_matrix_hadamard(X1, X2, G, *, variance, lengthscales, family="SquaredExponential", out=None):
    assert tuple(G.shape) == (X1.shape[0], X2.shape[0])
    R = torch.from_numpy(_k(X1, X2, variance, lengthscales, family) * _np(G))
    if out is None:
        return R
    out.copy_(R)
    return out


def kernel_matrix_combine(X1, X2, G, *, op, variance, lengthscales, family="SquaredExponential", diag_add=0.0, out=None):
    if op == "dr2":
        R = _dr2(X1, X1 if X2 is None else X2, variance, lengthscales, family) * _np(G)
        if X2 is None:
            np.fill_diagonal(R, 0.0)
    else:
        K = _k(X1, X1 if X2 is None else X2, variance, lengthscales, family)
        R = K * _np(G) if op == "mul" else K + _np(G)
        if X2 is None:
            R = R + diag_add * np.eye(R.shape[0])
    Rt = torch.from_numpy(R)
    if out is None:
        return Rt
    out.copy_(Rt)
    return out


def _invd(n, batch, mark):
    """Stand-in for the diagonal-block inverses: the emulation solves with L itself, the tensor only carries a marker
    (+1: belongs to L, -1: to L^T from transpose_factor) so that a wrong pairing is caught."""
    return torch.full((batch * (-(-n // NB)) * NB * NB,), mark, dtype=torch.float64)


def invd_alloc(n, batch=1):
    return _invd(n, batch, 0.0)


def _chol_info(K):
    """(L, info): info = j + 1 of the first non-positive pivot (LAPACK convention), L garbage from there on."""
    try:
        return np.linalg.cholesky(K), 0
    except np.linalg.LinAlgError:
        n = K.shape[0]
        for j in range(1, n + 1):
            try:
                np.linalg.cholesky(K[:j, :j])
            except np.linalg.LinAlgError:
                return np.full_like(K, np.nan), j
        return np.full_like(K, np.nan), n


def potrf_(T, n, *, zero_upper=False, invd=None, identity_rows=False):
    if identity_rows:
        assert T.dim() == 2 and T.shape[0] >= 2 * n
        T[T.shape[0] - n:] = torch.eye(n, dtype=T.dtype)
    if T.dim() == 3:
        infos = []
        for b in range(T.shape[0]):
            _, info = potrf_(T[b], n, zero_upper=zero_upper)
            infos.append(int(info[0]))
        return _invd(n, T.shape[0], +1.0), torch.tensor(infos, dtype=torch.int32)
    assert T.dim() == 2 and T.shape[1] == n and T.stride(1) == 1
    K = np.tril(_np(T[:n]))
    K = K + np.tril(K, -1).T          # only the lower triangle is read
    L, bad = _chol_info(K)
    if bad:
        return _invd(n, 1, +1.0), torch.tensor([bad], dtype=torch.int32)
    E = _np(T[n:])
    S = sla.solve_triangular(L, E.T, lower=True).T if E.shape[0] else E
    up = _np(T[:n]) * np.triu(np.ones((n, n)), 1)
    T[:n] = torch.from_numpy(L + (0.0 if zero_upper else up))
    T[n:] = torch.from_numpy(S)
    return _invd(n, 1, +1.0), torch.zeros(1, dtype=torch.int32)


def check_info(info, what="Cholesky"):
    from gpflow_amd._lib import GpkError
    bad = _np(info)
    if np.any(bad != 0):
        raise GpkError(f"{what} decomposition was not successful: non-positive pivot at column {int(bad[bad != 0][0]) - 1}")


def trtri_blocks(L):
    return _invd(L.shape[0], 1, +1.0)


def transpose_factor(L, invd):
    assert float(invd.reshape(-1)[0]) == 1.0
    return torch.tril(L).t().contiguous(), _invd(L.shape[0], 1, -1.0)


def trsm_(B, L, invd, *, trans=0):
    """trans=0: B <- B L^-T given (L, invd); trans=1: B <- B L^-1 given (LT, invdT)."""
    n = L.shape[0]
    if trans == 0:
        assert float(invd.reshape(-1)[0]) == 1.0, "trans=0 needs (L, invd)"
        Ll = np.tril(_np(L))
        B.copy_(torch.from_numpy(sla.solve_triangular(Ll, _np(B).T, lower=True).T))
    else:
        assert float(invd.reshape(-1)[0]) == -1.0, "trans=1 needs (LT, invdT) from transpose_factor"
        Ll = np.triu(_np(L)).T      # the argument is L^T (upper); only that triangle is read
        # B L^-1 = (L^-T B^T)^T
        B.copy_(torch.from_numpy(sla.solve_triangular(Ll.T, _np(B).T, lower=False).T))
    return B


def gemm_nt(A, B, *, alpha=1.0, beta=0.0, C=None, b_tri=0, c_lower=False, a_tri=0, k_split=False, zero_skipped=True):
    if k_split:   # batch entries = consecutive K chunks of one product; the structure statements are about the unsplit column index
        assert A.dim() == 3 and B.dim() == 3 and A.shape[0] == B.shape[0] and A.shape[2] % 16 == 0
    if a_tri:   # the hint must be TRUE: the device kernel skips the K range it declares zero
        A2 = _np(torch.cat(list(A), dim=1)) if k_split else _np(A if A.dim() == 2 else A[0])
        assert np.all((np.tril(A2, -1) if a_tri == 1 else np.triu(A2, 1)) == 0), "a_tri set on a matrix without that structure"
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
        if not c_lower or not zero_skipped:
            C.fill_(float("nan"))     # every entry must be written by the kernel (zero_skipped=False: skipped tiles stay uninitialised)
    C3 = C if C.dim() == 3 else C.unsqueeze(0)
    for z in range(batch):
        a = _np(A3[z if A3.shape[0] > 1 else 0])
        b = _np(B3[z if B3.shape[0] > 1 else 0]).copy()
        for n0 in range(0, n, NB):
            n1 = min(n0 + NB, n)
            kb, ke = 0, k
            koff = z * k if k_split else 0
            if b_tri == 1:
                kb = min(max(n0 - koff, 0) & ~15, k)
            elif b_tri == 2:
                ke = max(min(n0 + NB - koff, k), kb)
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
    R = Y.transpose(-1, -2).clone(memory_format=torch.contiguous_format)   # always new storage, like the kernel's output
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


def _noise(noise_variance, rows):
    """The (noise_variance, noise_rows) pair of the C-ABI: a constant, or one variance per row -> [rows, 1]."""
    if isinstance(noise_variance, (torch.Tensor, np.ndarray)) and int(np.prod(tuple(noise_variance.shape))) != 1:
        nv = _np(noise_variance).reshape(-1)
        assert nv.size == rows, (nv.size, rows)
        return nv[:, None]
    return float(noise_variance)


def diag_add_(A, v):
    """gpk_diag_add: A[i,i] += v[i] in place"""
    v = to_device(v).reshape(-1)
    n = min(A.shape[0], A.shape[1])
    assert v.numel() == n
    idx = torch.arange(n)
    A[idx, idx] += v
    return A


def gaussian_varexp_sum(Y, fmean, *, s0, ssq, knn, noise_variance, mean_const=0.0, s0_per_latent=False,
                        want_fvar=False):
    P = fmean.shape[1]
    knn = np.broadcast_to(np.atleast_1d(np.asarray(knn, dtype=np.float64)), (P,)) if np.size(knn) in (1, P) else None
    fv = np.tile(knn[None, :], (fmean.shape[0], 1)).astype(np.float64)
    if s0 is not None:
        fv = fv - (_np(s0).T if s0_per_latent else _np(s0)[:, None])
    if ssq is not None:
        fv = fv + _np(ssq).T
    nv = _noise(noise_variance, fmean.shape[0])
    ve = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(nv) \
        - 0.5 * ((_np(Y) - _np(fmean) - mean_const) ** 2 + fv) / nv
    return torch.tensor([ve.sum()], dtype=torch.float64), (torch.from_numpy(fv) if want_fvar else None)


def gauss_kl_white(q_mu, q_sqrt):
    M, P = q_mu.shape
    if q_sqrt.dim() == 2:      # q_diag: std-devs [M, P]
        s = _np(q_sqrt)
        kl = 0.5 * ((_np(q_mu) ** 2).sum() - M * P - np.log(s ** 2).sum() + (s * s).sum())
        return torch.tensor([kl], dtype=torch.float64)
    Lq = np.tril(_np(q_sqrt))
    kl = 0.5 * ((_np(q_mu) ** 2).sum() - M * P - np.log(np.diagonal(Lq, axis1=1, axis2=2) ** 2).sum() + (Lq * Lq).sum())
    return torch.tensor([kl], dtype=torch.float64)


def combine_parts(parts, *, alpha=1.0, lower=False, diag_scale=1.0, out=None):
    p3 = _np(parts if parts.dim() == 3 else parts.unsqueeze(0))
    if lower:   # entries above the diagonal may be NaN (unwritten tiles of a lower-only GEMM): never read
        p3 = np.where(np.tril(np.ones(p3.shape[1:], dtype=bool))[None], p3, 0.0)
    r = alpha * p3.sum(0)
    if lower:
        r = np.tril(r)
        r[np.diag_indices(min(r.shape))] *= diag_scale
    t = torch.from_numpy(r)
    if out is not None:
        out.copy_(t)
        return out
    return t


def sumsq(A, *, upper_only=False):
    a = _np(A)
    if upper_only:
        a = np.triu(a)
    return torch.tensor([(a * a).sum()], dtype=torch.float64)


def sum_log_diag(L):
    L3 = L if L.dim() == 3 else L.unsqueeze(0)
    return torch.from_numpy(np.log(np.diagonal(_np(L3), axis1=1, axis2=2)).sum(1))


def row_dot(A, B):
    return torch.from_numpy((_np(A) * _np(B)).sum(1))


def project(At, LqT):
    """ssq [P, rows] = sum_j (At Lq_p)[b, j]^2 with LqT[p] = tril(q_sqrt_p)^T (already triangular-clean)."""
    a = _np(At)
    return torch.from_numpy(np.stack([(((a[p] if a.ndim == 3 else a) @ _np(LqT[p]).T) ** 2).sum(1)
                                      for p in range(LqT.shape[0])]))


def gpr_lml(X, Y, *, variance, lengthscales, noise_variance, mean_const=0.0, family="SquaredExponential", ws=None):
    """The fused driver, emulated by the same chain of primitives it runs (potrf.hip: gpk_gpr_lml)."""
    n, P = Y.shape
    T = torch.empty((n + P, n), dtype=torch.float64)
    nv = _noise(noise_variance, n)
    kernel_matrix(X, None, variance=variance, lengthscales=lengthscales, family=family, diag_add=0.0 if isinstance(nv, np.ndarray) else nv,
                  lower_only=True, out=T[:n])
    if isinstance(nv, np.ndarray):
        diag_add_(T[:n], torch.from_numpy(nv[:, 0].copy()))
    T[n:] = (Y - mean_const).t()
    _, info = potrf_(T, n)
    if int(info[0]):
        return torch.full((1,), float("nan"), dtype=torch.float64), info
    lml = -0.5 * sumsq(T[n:])[0] - 0.5 * n * P * np.log(2 * np.pi) - P * sum_log_diag(T[:n])[0]
    return lml.reshape(1), info


def svgp_elbo_sep_workspace(m, rows, d, P):
    return torch.empty(1, dtype=torch.float64)


def svgp_elbo_shard_sep(Z, Xb, Yb, q_mu, q_sqrt, *, variances, lengthscales, families, noise_variance, jitter, mean_const=0.0,
                        ws=None, out=None, info=None):
    """gpk_svgp_elbo_shard_sep emulated by its chain of primitives (whitened; one kernel per latent; full q_sqrt)."""
    P = q_mu.shape[1]
    M, rows = Z.shape[-2], Xb.shape[0]
    ls = np.asarray(lengthscales, dtype=np.float64)
    T = torch.empty((P, M + rows, M), dtype=torch.float64)
    for p in range(P):
        Zp = Z if Z.dim() == 2 else Z[p]
        kw = dict(variance=float(variances[p]), lengthscales=ls[p], family=families[p])
        kernel_matrix(Zp, None, diag_add=jitter, lower_only=True, out=T[p, :M], **kw)
        if rows:
            kernel_matrix(Xb, Zp, out=T[p, M:], **kw)
    _, inf = potrf_(T, M)
    res = torch.zeros(2, dtype=torch.float64)
    if not bool((inf != 0).any()):
        s0 = torch.stack([row_stats(T[p, M:].contiguous(), V=q_mu[:, p:p + 1].contiguous())[0] for p in range(P)])
        fmean = torch.stack([row_stats(T[p, M:].contiguous(), V=q_mu[:, p:p + 1].contiguous())[1][:, 0] for p in range(P)], dim=1)
        ssq = project(T[:, M:], transpose(q_sqrt, mode=1))
        ve, _ = gaussian_varexp_sum(Yb, fmean.contiguous(), s0=s0, ssq=ssq, knn=[float(v) for v in variances],
                                    noise_variance=noise_variance, mean_const=mean_const, s0_per_latent=True)
        res[0] = ve[0]
        res[1] = gauss_kl_white(q_mu, q_sqrt)[0]
    if out is not None:
        out.copy_(res)
        res = out
    return res, inf


def svgp_elbo_workspace(m, rows, d, P, q_diag, whiten=True):
    return torch.empty(1, dtype=torch.float64)


def svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, *, variance, lengthscales, noise_variance, jitter, mean_const=0.0,
                    family="SquaredExponential", ws=None, out=None, info=None, whiten=True):
    """gpk_svgp_elbo_shard emulated by its own chain of primitives (shared kernel; whitened, or un-whitened on one trapezoid)."""
    M, rows, P = Z.shape[0], Xb.shape[0], q_mu.shape[1]
    kw = dict(variance=variance, lengthscales=lengthscales, family=family)
    if not whiten and q_sqrt.dim() == 2:
        # un-whitened, diagonal q_sqrt: [Kuu ; Kfu ; q_mu^T ; I] -> A^T, (Lm^-1 q_mu)^T, Lm^-T (potrf.hip, round 5)
        T = torch.empty((M + rows + P + M, M), dtype=torch.float64)
        kernel_matrix(Z, None, diag_add=jitter, lower_only=True, out=T[:M], **kw)
        if rows:
            kernel_matrix(Xb, Z, out=T[M:M + rows], **kw)
        T[M + rows:M + rows + P] = q_mu.t()
        _, inf = potrf_(T, M, identity_rows=True)
        res = torch.zeros(2, dtype=torch.float64)
        if int(inf[0]) == 0:
            At = T[M:M + rows].contiguous()
            LinvT = T[M + rows + P:].contiguous()
            A2 = gemm_nt(At, LinvT, b_tri=1) if rows else At
            s0 = row_stats(At)[0] if rows else torch.zeros(0, dtype=torch.float64)
            _, fmean, ssq = row_stats(A2, V=q_mu, W=q_sqrt, want_sumsq=False) if rows else (None, torch.zeros((0, P), dtype=torch.float64), torch.zeros((P, 0), dtype=torch.float64))
            ve, _ = gaussian_varexp_sum(Yb, fmean, s0=s0, ssq=ssq, knn=[variance], noise_variance=noise_variance,
                                        mean_const=mean_const)
            res[0] = ve[0]
            kinv = (np.triu(_np(LinvT)) ** 2).sum(1)
            w = _np(q_sqrt)
            res[1] = 0.5 * float((_np(T[M + rows:M + rows + P]) ** 2).sum()) + 0.5 * float((kinv[:, None] * w ** 2).sum()) \
                - 0.5 * float(np.log(w ** 2).sum()) - 0.5 * M * P + P * float(np.log(np.diagonal(_np(T[:M]))).sum())
        if out is not None:
            out.copy_(res)
            res = out
        return res, inf
    if not whiten:
        T = torch.empty((M + rows + P + P * M, M), dtype=torch.float64)
        kernel_matrix(Z, None, diag_add=jitter, lower_only=True, out=T[:M], **kw)
        if rows:
            kernel_matrix(Xb, Z, out=T[M:M + rows], **kw)
        T[M + rows:M + rows + P] = q_mu.t()
        T[M + rows + P:] = transpose(q_sqrt, mode=1).reshape(P * M, M)
        _, inf = potrf_(T, M)
        res = torch.zeros(2, dtype=torch.float64)
        if int(inf[0]) == 0:
            At = T[M:M + rows].contiguous()
            V = T[M + rows:M + rows + P].t().contiguous()
            GT = T[M + rows + P:].reshape(P, M, M).contiguous()
            s0, fmean, _ = row_stats(At, V=V)
            ssq = project(At, GT)
            ve, _ = gaussian_varexp_sum(Yb, fmean, s0=s0, ssq=ssq, knn=[variance], noise_variance=noise_variance,
                                        mean_const=mean_const)
            res[0] = ve[0]
            Lq = np.tril(_np(q_sqrt))
            res[1] = 0.5 * float((_np(T[M + rows:M + rows + P]) ** 2).sum()) + 0.5 * float((_np(GT) ** 2).sum()) - 0.5 * M * P \
                - 0.5 * float(np.log(np.diagonal(Lq, axis1=1, axis2=2) ** 2).sum()) \
                + P * float(np.log(np.diagonal(_np(T[:M]))).sum())
        if out is not None:
            out.copy_(res)
            res = out
        return res, inf
    T = torch.empty((M + rows, M), dtype=torch.float64)
    kernel_matrix(Z, None, diag_add=jitter, lower_only=True, out=T[:M], **kw)
    if rows:
        kernel_matrix(Xb, Z, out=T[M:], **kw)
    _, inf = potrf_(T, M)
    res = torch.zeros(2, dtype=torch.float64)
    if int(inf[0]) == 0:
        At = T[M:]
        if q_sqrt.dim() == 2:
            s0, fmean, wsq = row_stats(At, V=q_mu, W=q_sqrt)
            ssq = wsq
        else:
            s0, fmean, _ = row_stats(At, V=q_mu)
            ssq = project(At.contiguous(), transpose(q_sqrt, mode=1))
        ve, _ = gaussian_varexp_sum(Yb, fmean, s0=s0, ssq=ssq, knn=[variance], noise_variance=noise_variance,
                                    mean_const=mean_const)
        res[0] = ve[0]
        res[1] = gauss_kl_white(q_mu, q_sqrt)[0]
    if out is not None:
        out.copy_(res)
        res = out
    return res, inf


# ---- reverse-pass glue (include/gpk.h: gpk_moment_rows, gpk_stationary_adjoint_tail, gpk_adam_step, gpk_symmetrize) ----
def moment_rows(B):
    Bt = B.t()
    return torch.cat([torch.ones((1, B.shape[0]), dtype=torch.float64), Bt, Bt * Bt], 0).contiguous()


def stationary_adjoint_tail(R, A, ls, *, variance, symmetric, sum_kbar_k=None, into=None, dvar_add=0.0):
    D = A.shape[1]
    rs, GB, GB2 = R[:, 0:1], R[:, 1:1 + D], R[:, 1 + D:]
    T = GB - A * rs
    if symmetric:
        Abar = T * (2.0 / (ls * ls))
        dls = -(A * Abar).sum(0) / ls
    else:
        Abar = T / (ls * ls)
        dls = (GB2 - A * (GB + T)).sum(0) / ls ** 3
    dvar = (rs.sum() if sum_kbar_k is None else sum_kbar_k.reshape(())) / variance + dvar_add
    if into is not None:
        small, acc = into
        small[0:1] += dvar.reshape(1)
        small[1:] += dls
        acc += Abar
        return small[0:1], small[1:], acc
    small = torch.cat([dvar.reshape(1), dls])
    return small[0:1], small[1:], Abar


def adam_step_(p, g, m, v, *, beta1, beta2, epsilon, step, maximise=False):
    gg = -g if maximise else g
    m.mul_(beta1).add_(gg, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1.0 - beta2)
    p.addcdiv_(m, v.sqrt().add_(epsilon), value=-step)
    return p


def symmetrize_(S):
    S.copy_(0.5 * (S + S.t()))
    return S


def lowrank_axpy(alpha, X, U, V):
    return alpha * X + U @ V.t()
