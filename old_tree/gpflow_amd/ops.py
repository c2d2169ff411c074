"""Device primitives: thin, typed wrappers from torch fp64 HIP tensors to the C-ABI of libgpk.so.

torch is plumbing here (device memory, the current HIP stream, dlpack-style pointers); all arithmetic
happens in the hand-written kernels.  Every function validates device/dtype/layout and raises --
nothing silently falls back to torch math or to the CPU.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

NB = 128
KERNEL_FAMILIES = {"SquaredExponential": 0, "Matern12": 1, "Matern32": 2, "Matern52": 3}
MAX_D = 64


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.GpkError("gpflow_amd needs a HIP device (torch.cuda.is_available() is False); "
                            "there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def to_device(x, dtype=torch.float64) -> torch.Tensor:
    """NumPy / torch / scalar -> contiguous fp64 tensor on the current HIP device."""
    if isinstance(x, torch.Tensor):
        t = x.to(device=device(), dtype=dtype)
    else:
        t = torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=dtype, device=device())
    return t.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, ndim: Optional[int] = None) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.GpkError(f"{name}: tensor must live on the HIP device (got {t.device})")
    if t.dtype != torch.float64:
        raise _lib.GpkError(f"{name}: dtype must be float64 (got {t.dtype})")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")


def _rowmajor(t: torch.Tensor, name: str) -> int:
    """Leading dimension of a 2-D row-major view (last stride 1)."""
    if t.dim() != 2:
        raise ValueError(f"{name}: expected a matrix, got shape {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise _lib.GpkError(f"{name}: last dimension must be contiguous")
    if t.shape[0] > 1:
        return int(t.stride(0))
    return int(max(t.shape[1], 1))


INFO_HANDOFF_TIMEOUT = 2 ** 31 - 1   # include/gpk.h: status of a factorisation whose internal hand-off never arrived


def _ws(nbytes: int) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 8) + 7) // 8, dtype=torch.float64, device=device())


def _ls_host(lengthscales, d: int):
    ls = np.atleast_1d(np.asarray(lengthscales, dtype=np.float64))
    ard = ls.size > 1
    if ard and ls.size != d:
        raise ValueError(f"lengthscales has {ls.size} entries, input dimension is {d}")
    return _lib.host_doubles(ls.tolist()), int(ard)


def _noise_args(noise_variance, rows: int):
    """(scalar, device pointer | None, keep-alive) for the (noise_variance, noise_rows) pair of the C-ABI: a float is the constant
    noise of a homoskedastic Gaussian likelihood; a tensor / array with `rows` entries is one variance per data row
    (Gaussian(variance=Function | scale=Function), likelihoods/scalar_continuous.py:92-111)."""
    if isinstance(noise_variance, (torch.Tensor, np.ndarray)) and int(np.prod(tuple(noise_variance.shape))) != 1:
        nv = to_device(noise_variance).reshape(-1).contiguous()
        if nv.numel() != rows:
            raise ValueError(f"per-row noise variances: expected {rows} entries, got {nv.numel()}")
        return 1.0, nv.data_ptr(), nv
    return float(noise_variance), None, None


# ------------------------------------------------------------------------------------------------
def kernel_matrix(X1: torch.Tensor, X2: Optional[torch.Tensor], *, variance: float, lengthscales,
                  family: str = "SquaredExponential", diag_add: float = 0.0, lower_only: bool = False,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K(X1, X2) (or K(X1, X1) + diag_add I when X2 is None) -> [n1, n2]."""
    lib = _lib.load()
    _chk(X1, "X1", 2)
    n1, d = X1.shape
    if d < 1 or d > MAX_D:
        raise ValueError(f"input dimension {d} outside [1, {MAX_D}]")
    if X2 is not None:
        _chk(X2, "X2", 2)
        if X2.shape[1] != d:
            raise ValueError("X1 and X2 must have the same number of columns")
        n2 = X2.shape[0]
    else:
        n2 = n1
    if out is None:
        out = torch.empty((n1, n2), dtype=torch.float64, device=X1.device)
        if lower_only:
            out.zero_()
    _chk(out, "out", 2)
    ls, ard = _ls_host(lengthscales, d)
    rc = lib.gpk_kernel_matrix(_stream(), KERNEL_FAMILIES[family], X1.data_ptr(), n1, _rowmajor(X1, "X1"),
                               X2.data_ptr() if X2 is not None else None, n2,
                               _rowmajor(X2, "X2") if X2 is not None else 0, d, ls, ard,
                               float(variance), float(diag_add), int(lower_only), out.data_ptr(),
                               _rowmajor(out, "out"))
    _lib.check(rc, "gpk_kernel_matrix")
    return out


def kernel_matrix_hadamard(X1: torch.Tensor, X2: torch.Tensor, G: torch.Tensor, *, variance: float, lengthscales,
                           family: str = "SquaredExponential", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """G .* K(X1, X2) with K recomputed on the fly -> [n1, n2] (the elementwise factor of the kernel backward)."""
    lib = _lib.load()
    _chk(X1, "X1", 2)
    _chk(X2, "X2", 2)
    _chk(G, "G", 2)
    n1, d = X1.shape
    n2 = X2.shape[0]
    if X2.shape[1] != d or tuple(G.shape) != (n1, n2):
        raise ValueError("inconsistent shapes")
    if d < 1 or d > MAX_D:
        raise ValueError(f"input dimension {d} outside [1, {MAX_D}]")
    if out is None:
        out = torch.empty((n1, n2), dtype=torch.float64, device=X1.device)
    _chk(out, "out", 2)
    ls, ard = _ls_host(lengthscales, d)
    rc = lib.gpk_kernel_matrix_hadamard(_stream(), KERNEL_FAMILIES[family], X1.data_ptr(), n1, _rowmajor(X1, "X1"),
                                        X2.data_ptr(), n2, _rowmajor(X2, "X2"), d, ls, ard, float(variance),
                                        G.data_ptr(), _rowmajor(G, "G"), out.data_ptr(), _rowmajor(out, "out"))
    _lib.check(rc, "gpk_kernel_matrix_hadamard")
    return out


def kernel_matrix_combine(X1: torch.Tensor, X2: Optional[torch.Tensor], G: torch.Tensor, *, op: str, variance: float,
                          lengthscales, family: str = "SquaredExponential", diag_add: float = 0.0,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = G .* K(X1, X2) (op "mul"), G + K(X1, X2) (op "add") or G .* (-2 dK/dr2)(X1, X2) (op "dr2": the factor of the
    lengthscale / input gradients of a stationary kernel; r2 = scaled squared distance), K recomputed on the fly; `out`
    may be G itself.  X2 None: K(X1, X1); `diag_add` goes onto the diagonal of the combined result ("mul" / "add"), and
    "dr2" writes exact zeros on the diagonal."""
    lib = _lib.load()
    _chk(X1, "X1", 2)
    _chk(G, "G", 2)
    n1, d = X1.shape
    if X2 is not None:
        _chk(X2, "X2", 2)
        if X2.shape[1] != d:
            raise ValueError("X1 and X2 must have the same number of columns")
    n2 = X2.shape[0] if X2 is not None else n1
    if tuple(G.shape) != (n1, n2):
        raise ValueError("inconsistent shapes")
    if d < 1 or d > MAX_D:
        raise ValueError(f"input dimension {d} outside [1, {MAX_D}]")
    if out is None:
        out = torch.empty((n1, n2), dtype=torch.float64, device=X1.device)
    _chk(out, "out", 2)
    ls, ard = _ls_host(lengthscales, d)
    rc = lib.gpk_kernel_matrix_combine(_stream(), KERNEL_FAMILIES[family], {"mul": 1, "add": 2, "dr2": 3}[op], X1.data_ptr(), n1,
                                       _rowmajor(X1, "X1"), X2.data_ptr() if X2 is not None else None, n2,
                                       _rowmajor(X2, "X2") if X2 is not None else 0, d, ls, ard, float(variance),
                                       float(diag_add), G.data_ptr(), _rowmajor(G, "G"), out.data_ptr(),
                                       _rowmajor(out, "out"))
    _lib.check(rc, "gpk_kernel_matrix_combine")
    return out


def diag_add_(A: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """A[i,i] += v[i] in place (add_noise_cov with a per-row variance, utilities/model_utils.py:33-38) -- gpk_diag_add."""
    lib = _lib.load()
    _chk(A, "A", 2)
    v = to_device(v).reshape(-1).contiguous()
    n = min(A.shape[0], A.shape[1])
    if v.numel() != n:
        raise ValueError("diag_add_: one value per diagonal entry")
    _lib.check(lib.gpk_diag_add(_stream(), A.data_ptr(), n, _rowmajor(A, "A"), v.data_ptr()), "gpk_diag_add")
    return A


def invd_alloc(n: int, batch: int = 1) -> torch.Tensor:
    lib = _lib.load()
    return torch.empty(int(lib.gpk_invd_elems(n, batch)), dtype=torch.float64, device=device())


def potrf_(T: torch.Tensor, n: int, *, zero_upper: bool = False,
           invd: Optional[torch.Tensor] = None, identity_rows: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """In-place trapezoidal Cholesky of T [(n+extra), n] or batched [b, (n+extra), n].
    Returns (invd, info) -- info is a device int32 tensor (0 ok, j+1 first bad pivot).

    identity_rows=True (2-D only): the LAST n rows of T are overwritten with the identity by the library and come back
    as L^-T at a third of the cost of n dense rows (gpk_potrf_inv) -- the caller leaves them uninitialised."""
    lib = _lib.load()
    _chk(T, "T")
    if T.dim() == 2:
        batch, rows, cols, stride = 1, T.shape[0], T.shape[1], 0
        lda = _rowmajor(T, "T")
    elif T.dim() == 3:
        batch, rows, cols = T.shape
        stride = int(T.stride(0))
        lda = _rowmajor(T[0], "T")
    else:
        raise ValueError("T must be 2-D or 3-D")
    if cols != n or rows < n:
        raise ValueError(f"T has shape {tuple(T.shape)}, expected [.., n+extra, n] with n={n}")
    if invd is None:
        invd = invd_alloc(n, batch)
    info = torch.zeros(batch, dtype=torch.int32, device=T.device)
    if identity_rows:
        if T.dim() != 2 or rows < 2 * n:
            raise ValueError("identity_rows needs a 2-D T with at least 2 n rows")
        rc = lib.gpk_potrf_inv(_stream(), T.data_ptr(), n, rows - 2 * n, lda, invd.data_ptr(), int(zero_upper),
                               info.data_ptr())
        _lib.check(rc, "gpk_potrf_inv")
        return invd, info
    rc = lib.gpk_potrf(_stream(), T.data_ptr(), n, rows - n, lda, batch, stride, invd.data_ptr(),
                       int(zero_upper), info.data_ptr())
    _lib.check(rc, "gpk_potrf")
    return invd, info


def check_info(info: torch.Tensor, what: str = "Cholesky") -> None:
    """Synchronising check of the factorisation status (the reference raises InvalidArgumentError
    'Cholesky decomposition was not successful' from tf.linalg.cholesky on CPU)."""
    bad = info.cpu().numpy()
    if np.any(bad != 0):
        j = int(bad[np.nonzero(bad)[0][0]])
        if j >= INFO_HANDOFF_TIMEOUT:   # include/gpk.h, "info": an internal stream hand-off of the factorisation timed out
            raise _lib.GpkError(f"{what} decomposition failed: an internal stream hand-off timed out (status INT_MAX); "
                                "the result is undefined")
        raise _lib.GpkError(f"{what} decomposition was not successful: non-positive pivot at column {j - 1}")


def trtri_blocks(L: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _chk(L, "L", 2)
    n = L.shape[0]
    invd = invd_alloc(n, 1)
    rc = lib.gpk_trtri_blocks(_stream(), L.data_ptr(), n, _rowmajor(L, "L"), 1, 0, invd.data_ptr())
    _lib.check(rc, "gpk_trtri_blocks")
    return invd


def transpose_factor(L: torch.Tensor, invd: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    _chk(L, "L", 2)
    n = L.shape[0]
    LT = torch.empty((n, n), dtype=torch.float64, device=L.device)
    invdT = torch.empty_like(invd)
    rc = lib.gpk_transpose_factor(_stream(), L.data_ptr(), _rowmajor(L, "L"), invd.data_ptr(), n,
                                  LT.data_ptr(), n, invdT.data_ptr())
    _lib.check(rc, "gpk_transpose_factor")
    return LT, invdT


def trsm_(B: torch.Tensor, L: torch.Tensor, invd: torch.Tensor, *, trans: int = 0) -> torch.Tensor:
    """In place: trans=0  B <- B L^-T (pass L, invd);  trans=1  B <- B L^-1 (pass LT, invdT)."""
    lib = _lib.load()
    _chk(B, "B", 2)
    _chk(L, "L", 2)
    n = L.shape[0]
    if B.shape[1] != n:
        raise ValueError(f"B has {B.shape[1]} columns, factor is {n}x{n}")
    rc = lib.gpk_trsm(_stream(), int(trans), L.data_ptr(), _rowmajor(L, "L"), invd.data_ptr(), n,
                      B.data_ptr(), B.shape[0], _rowmajor(B, "B"), 1, 0, 0)
    _lib.check(rc, "gpk_trsm")
    return B


def gemm_nt(A: torch.Tensor, B: torch.Tensor, *, alpha: float = 1.0, beta: float = 0.0,
            C: Optional[torch.Tensor] = None, b_tri: int = 0, c_lower: bool = False, a_tri: int = 0,
            k_split: bool = False, zero_skipped: bool = True) -> torch.Tensor:
    """C = alpha A B^T + beta C.  A [m,k] or [b,m,k]; B [n,k] or [b,n,k].  b_tri: 1 B upper (B[j,kk] = 0 for kk < j),
    2 B lower; a_tri: the same statement about A (1 upper, 2 lower) -- zeros must be stored; only the K ranges shrink.
    k_split: the batch entries are consecutive K chunks of ONE triangular product (strided views of its operands); b_tri / a_tri
    then refer to the unsplit column index and the caller sums the partial products (combine_parts).
    c_lower without C: the tiles above the diagonal, which the kernel skips, are zero-filled first -- unless zero_skipped=False (a caller
    that never reads them, e.g. combine_parts(lower=True) behind a split-K product: the fill of 8 x 2048^2 partials was 72 us)."""
    lib = _lib.load()
    _chk(A, "A")
    _chk(B, "B")
    batched = A.dim() == 3 or B.dim() == 3
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    B3 = B if B.dim() == 3 else B.unsqueeze(0)
    batch = max(A3.shape[0], B3.shape[0])
    m, k = A3.shape[1], A3.shape[2]
    n = B3.shape[1]
    if B3.shape[2] != k:
        raise ValueError("inner dimensions differ")
    if C is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs C")
        C = torch.empty((batch, m, n) if batched else (m, n), dtype=torch.float64, device=A.device)
        if c_lower and zero_skipped:
            C.zero_()
    C3 = C if C.dim() == 3 else C.unsqueeze(0)
    sA = int(A3.stride(0)) if A3.shape[0] > 1 else 0
    sB = int(B3.stride(0)) if B3.shape[0] > 1 else 0
    sC = int(C3.stride(0)) if C3.shape[0] > 1 else 0
    rc = lib.gpk_gemm_nt(_stream(), m, n, k, float(alpha), A3.data_ptr(), _rowmajor(A3[0], "A"),
                         B3.data_ptr(), _rowmajor(B3[0], "B"), float(beta), C3.data_ptr(),
                         _rowmajor(C3[0], "C"), int(b_tri) | (int(a_tri) << 4) | (0x100 if k_split else 0), int(c_lower), batch, sA, sB, sC)
    _lib.check(rc, "gpk_gemm_nt")
    return C


def transpose(X: torch.Tensor, *, mode: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[.., j, i] = X[.., i, j]; mode 1 keeps only the lower triangle of X, 2 only the upper."""
    lib = _lib.load()
    _chk(X, "X")
    X3 = X if X.dim() == 3 else X.unsqueeze(0)
    b, r, c = X3.shape
    if out is None:
        out = torch.empty((b, c, r) if X.dim() == 3 else (c, r), dtype=torch.float64, device=X.device)
    O3 = out if out.dim() == 3 else out.unsqueeze(0)
    rc = lib.gpk_transpose(_stream(), X3.data_ptr(), r, c, _rowmajor(X3[0], "X"), O3.data_ptr(),
                           _rowmajor(O3[0], "out"), int(mode), b,
                           int(X3.stride(0)) if b > 1 else 0, int(O3.stride(0)) if b > 1 else 0)
    _lib.check(rc, "gpk_transpose")
    return out


def row_stats(At: torch.Tensor, *, V: Optional[torch.Tensor] = None, W: Optional[torch.Tensor] = None,
              want_sumsq: bool = True):
    """(sumsq [rows], mv [rows,P] = At V, wsq [P,rows] = sum_k (At W)^2) -- any may be None."""
    lib = _lib.load()
    _chk(At, "At", 2)
    rows, m = At.shape
    P = 0
    for name, t in (("V", V), ("W", W)):
        if t is not None:
            _chk(t, name, 2)
            if t.shape[0] != m or not t.is_contiguous():
                raise ValueError(f"{name} must be contiguous [m, P]")
            P = t.shape[1]
    sumsq = torch.empty(rows, dtype=torch.float64, device=At.device) if want_sumsq else None
    mv = torch.empty((rows, P), dtype=torch.float64, device=At.device) if V is not None else None
    wsq = torch.empty((P, rows), dtype=torch.float64, device=At.device) if W is not None else None
    rc = lib.gpk_row_stats(_stream(), At.data_ptr(), rows, m, _rowmajor(At, "At"),
                           V.data_ptr() if V is not None else None,
                           W.data_ptr() if W is not None else None, P, 1.0, 0.0,
                           sumsq.data_ptr() if sumsq is not None else None,
                           mv.data_ptr() if mv is not None else None,
                           wsq.data_ptr() if wsq is not None else None)
    _lib.check(rc, "gpk_row_stats")
    return sumsq, mv, wsq


def row_dot(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """out[i] = sum_j A[i,j] B[i,j]"""
    lib = _lib.load()
    _chk(A, "A", 2)
    _chk(B, "B", 2)
    if A.shape != B.shape:
        raise ValueError("shape mismatch")
    out = torch.empty(A.shape[0], dtype=torch.float64, device=A.device)
    rc = lib.gpk_row_dot(_stream(), A.data_ptr(), _rowmajor(A, "A"), B.data_ptr(), _rowmajor(B, "B"),
                         A.shape[0], A.shape[1], 1.0, 0.0, out.data_ptr())
    _lib.check(rc, "gpk_row_dot")
    return out


def project(At: torch.Tensor, LqT: torch.Tensor) -> torch.Tensor:
    """ssq [P, rows] = sum_j (At_p Lq_p)[b, j]^2 with LqT [P, m, m] = tril(q_sqrt_p)^T.  At [rows, m]: one matrix shared by
    the P latents; At [P, rows, m] (any batch stride, unit column stride): one per latent (SeparateIndependent)."""
    lib = _lib.load()
    _chk(LqT, "LqT", 3)
    P = LqT.shape[0]
    if At.dim() == 3:
        if At.dtype != torch.float64 or At.shape[0] != P or At.stride(2) != 1:
            raise ValueError("batched At must be float64 [P, rows, m] with unit column stride")
        rows, m = At.shape[1], At.shape[2]
        ldat, stride_at = At.stride(1), At.stride(0)
    else:
        _chk(At, "At", 2)
        rows, m = At.shape
        ldat, stride_at = _rowmajor(At, "At"), 0
    if LqT.shape[1] != m or LqT.shape[2] != m or not LqT.is_contiguous():
        raise ValueError("LqT must be contiguous [P, m, m]")
    ssq = torch.empty((P, rows), dtype=torch.float64, device=At.device)
    nbytes = int(lib.gpk_project_workspace_bytes(rows, m, P))
    ws = _ws(nbytes)
    rc = lib.gpk_project_batched(_stream(), At.data_ptr(), rows, m, ldat, stride_at, LqT.data_ptr(), m, P,
                                 ssq.data_ptr(), ws.data_ptr(), ws.numel() * 8)
    _lib.check(rc, "gpk_project_batched")
    return ssq


def gaussian_varexp_sum(Y: torch.Tensor, fmean: torch.Tensor, *, s0: Optional[torch.Tensor],
                        ssq: Optional[torch.Tensor], knn: Sequence[float], noise_variance,
                        mean_const: float = 0.0, s0_per_latent: bool = False, want_fvar: bool = False):
    """Sum over rows/outputs of the Gaussian variational expectations; returns (scalar tensor, fvar|None).
    noise_variance: a float, or one variance per row [rows] (heteroskedastic likelihood)."""
    lib = _lib.load()
    _chk(Y, "Y", 2)
    _chk(fmean, "fmean", 2)
    rows, P = fmean.shape
    if not fmean.is_contiguous():
        raise ValueError("fmean must be contiguous")
    out = torch.empty(1, dtype=torch.float64, device=Y.device)
    fvar = torch.empty((rows, P), dtype=torch.float64, device=Y.device) if want_fvar else None
    ws = _ws(int(lib.gpk_reduce_workspace_bytes(rows)))
    knn = list(np.atleast_1d(np.asarray(knn, dtype=np.float64)))
    per = int(len(knn) > 1)
    nv, nv_rows, _keep = _noise_args(noise_variance, rows)
    rc = lib.gpk_gaussian_varexp_sum(_stream(), Y.data_ptr(), _rowmajor(Y, "Y"), fmean.data_ptr(), rows, P,
                                     s0.data_ptr() if s0 is not None else None, int(s0_per_latent),
                                     ssq.data_ptr() if ssq is not None else None,
                                     _lib.host_doubles(knn), per, nv, nv_rows, float(mean_const),
                                     fvar.data_ptr() if fvar is not None else None, out.data_ptr(),
                                     ws.data_ptr(), ws.numel() * 8)
    _lib.check(rc, "gpk_gaussian_varexp_sum")
    return out, fvar


def gauss_kl_white(q_mu: torch.Tensor, q_sqrt: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _chk(q_mu, "q_mu", 2)
    _chk(q_sqrt, "q_sqrt")
    m, P = q_mu.shape
    q_diag = int(q_sqrt.dim() == 2)
    if not (q_mu.is_contiguous() and q_sqrt.is_contiguous()):
        raise ValueError("q_mu / q_sqrt must be contiguous")
    out = torch.empty(1, dtype=torch.float64, device=q_mu.device)
    ws = _ws(int(lib.gpk_reduce_workspace_bytes(m)))
    rc = lib.gpk_gauss_kl_white(_stream(), q_mu.data_ptr(), q_sqrt.data_ptr(), m, P, q_diag,
                                out.data_ptr(), ws.data_ptr(), ws.numel() * 8)
    _lib.check(rc, "gpk_gauss_kl_white")
    return out


def sum_log_diag(L: torch.Tensor) -> torch.Tensor:
    """sum_i log L[i,i] for L [n,>=n] or batched [b,n,>=n] -> [b]."""
    lib = _lib.load()
    _chk(L, "L")
    L3 = L if L.dim() == 3 else L.unsqueeze(0)
    b = L3.shape[0]
    n = min(L3.shape[1], L3.shape[2])
    out = torch.empty(b, dtype=torch.float64, device=L.device)
    rc = lib.gpk_sum_log_diag(_stream(), L3.data_ptr(), n, _rowmajor(L3[0], "L"), b,
                              int(L3.stride(0)) if b > 1 else 0, out.data_ptr())
    _lib.check(rc, "gpk_sum_log_diag")
    return out


def combine_parts(parts: torch.Tensor, *, alpha: float = 1.0, lower: bool = False, diag_scale: float = 1.0,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """alpha * sum_p parts[p] for parts [np, m, n] (or [m, n]: one part), summed in the order p = 0, 1, ...; with
    lower=True the result is lower-triangular (zeros above the diagonal, which is never read -- a lower-only GEMM leaves
    those tiles unwritten) and its diagonal is multiplied by diag_scale (0.5: the Phi of the Cholesky adjoint)."""
    lib = _lib.load()
    _chk(parts, "parts")
    p3 = parts if parts.dim() == 3 else parts.unsqueeze(0)
    if p3.dim() != 3:
        raise ValueError("parts must be [np, m, n] or [m, n]")
    npart, m, n = p3.shape
    ldp = _rowmajor(p3[0], "parts")
    stride = int(p3.stride(0)) if npart > 1 else 0
    if out is None:
        out = torch.empty((m, n), dtype=torch.float64, device=parts.device)
    _chk(out, "out", 2)
    rc = lib.gpk_combine_parts(_stream(), p3.data_ptr(), npart, stride, m, n, ldp, float(alpha), int(lower),
                               float(diag_scale), out.data_ptr(), _rowmajor(out, "out"))
    _lib.check(rc, "gpk_combine_parts")
    return out


def sumsq(A: torch.Tensor, *, upper_only: bool = False) -> torch.Tensor:
    lib = _lib.load()
    _chk(A, "A", 2)
    out = torch.empty(1, dtype=torch.float64, device=A.device)
    ws = _ws(int(lib.gpk_reduce_workspace_bytes(A.shape[0])))
    rc = lib.gpk_sumsq(_stream(), A.data_ptr(), A.shape[0], A.shape[1], _rowmajor(A, "A"), int(upper_only),
                       out.data_ptr(), ws.data_ptr(), ws.numel() * 8)
    _lib.check(rc, "gpk_sumsq")
    return out


# ------------------------------------------------------------------------------------------------ reverse-pass glue
def moment_rows(B: torch.Tensor) -> torch.Tensor:
    """[1; B^T; (B^T)^2] as [1 + 2 D, n2] for B [n2, D]: the right-hand side of G [1, B, B^2] (gradients.stationary_kernel_adjoint)."""
    lib = _lib.load()
    _chk(B, "B", 2)
    n2, d = B.shape
    Vt = torch.empty((1 + 2 * d, n2), dtype=torch.float64, device=B.device)
    rc = lib.gpk_moment_rows(_stream(), B.data_ptr(), _rowmajor(B, "B"), n2, d, Vt.data_ptr(), n2)
    _lib.check(rc, "gpk_moment_rows")
    return Vt


def stationary_adjoint_tail(R: torch.Tensor, A: torch.Tensor, ls: torch.Tensor, *, variance: float, symmetric: bool,
                            sum_kbar_k: Optional[torch.Tensor] = None, into: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                            dvar_add: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(d/dvariance [1], d/dlengthscales [D], A_bar [n1, D]) from R [n1, 1 + 2 D] = G [1, B, B^2], the first kernel argument A
    [n1, D] and the lengthscales as a [D] device tensor (include/gpk.h: gpk_stationary_adjoint_tail) -- one launch.
    into = (small [1 + D], A_bar) of an earlier call: the results are ADDED to them (and views of them returned); dvar_add is added
    to d/dvariance."""
    lib = _lib.load()
    _chk(R, "R", 2)
    _chk(A, "A", 2)
    _chk(ls, "ls", 1)
    n1, d = A.shape
    if R.shape != (n1, 1 + 2 * d) or ls.shape[0] != d or not ls.is_contiguous():
        raise ValueError("stationary_adjoint_tail: R must be [n1, 1 + 2 D], ls a contiguous [D]")
    if sum_kbar_k is not None:
        _chk(sum_kbar_k, "sum_kbar_k")
    if into is None:
        Abar = torch.empty((n1, d), dtype=torch.float64, device=A.device)
        small = torch.empty(1 + d, dtype=torch.float64, device=A.device)
    else:
        small, Abar = into
        _chk(small, "into[0]", 1)
        _chk(Abar, "into[1]", 2)
        if small.shape[0] != 1 + d or tuple(Abar.shape) != (n1, d) or not small.is_contiguous():
            raise ValueError("stationary_adjoint_tail: into must be (small [1 + D], A_bar [n1, D])")
    rc = lib.gpk_stationary_adjoint_tail(_stream(), R.data_ptr(), _rowmajor(R, "R"), A.data_ptr(), _rowmajor(A, "A"), n1, d,
                                         ls.data_ptr(), float(variance), int(bool(symmetric)),
                                         None if sum_kbar_k is None else sum_kbar_k.data_ptr(), Abar.data_ptr(), _rowmajor(Abar, "A_bar"),
                                         small.data_ptr(), int(into is not None), float(dvar_add))
    _lib.check(rc, "gpk_stationary_adjoint_tail")
    return small[0:1], small[1:], Abar


def adam_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, beta1: float, beta2: float, epsilon: float,
               step: float, maximise: bool = False) -> torch.Tensor:
    """tf.keras Adam on one contiguous variable, in place (p, m, v): one launch instead of seven (include/gpk.h: gpk_adam_step).
    step = lr sqrt(1 - beta2^t) / (1 - beta1^t); maximise=True takes g as the gradient of the quantity to MAXIMISE."""
    lib = _lib.load()
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, name)
        if not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError(f"adam_step_: {name} must be contiguous and of the variable's size")
    rc = lib.gpk_adam_step(_stream(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(beta1), float(beta2),
                           float(epsilon), float(step), int(bool(maximise)))
    _lib.check(rc, "gpk_adam_step")
    return p


def lowrank_axpy(alpha: float, X: torch.Tensor, U: torch.Tensor, V: torch.Tensor) -> torch.Tensor:
    """alpha X + U V^T for X [m, n], thin U [m, k], V [n, k], k <= 16, as a fresh tensor: one pass over X (gpk_lowrank_axpy)."""
    lib = _lib.load()
    _chk(X, "X", 2)
    _chk(U, "U", 2)
    _chk(V, "V", 2)
    m, n = X.shape
    k = U.shape[1]
    if U.shape[0] != m or tuple(V.shape) != (n, k) or not 0 < k <= 16:
        raise ValueError("lowrank_axpy: X [m, n], U [m, k], V [n, k], k <= 16")
    out = torch.empty((m, n), dtype=torch.float64, device=X.device)
    rc = lib.gpk_lowrank_axpy(_stream(), float(alpha), X.data_ptr(), _rowmajor(X, "X"), U.data_ptr(), _rowmajor(U, "U"), V.data_ptr(),
                              _rowmajor(V, "V"), m, n, k, out.data_ptr(), n)
    _lib.check(rc, "gpk_lowrank_axpy")
    return out


def symmetrize_(S: torch.Tensor) -> torch.Tensor:
    """S = (S + S^T) / 2 in place for a square S (one launch)."""
    lib = _lib.load()
    _chk(S, "S", 2)
    if S.shape[0] != S.shape[1]:
        raise ValueError("symmetrize_: square matrix expected")
    rc = lib.gpk_symmetrize(_stream(), S.data_ptr(), S.shape[0], _rowmajor(S, "S"))
    _lib.check(rc, "gpk_symmetrize")
    return S


# ------------------------------------------------------------------------------------------------ fused
def gpr_lml(X: torch.Tensor, Y: torch.Tensor, *, variance: float, lengthscales, noise_variance,
            mean_const: float = 0.0, family: str = "SquaredExponential",
            ws: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(LML scalar tensor, info) -- gpk_gpr_lml.  noise_variance: a float, or one variance per data row [n]."""
    lib = _lib.load()
    _chk(X, "X", 2)
    _chk(Y, "Y", 2)
    n, d = X.shape
    P = Y.shape[1]
    if Y.shape[0] != n:
        raise ValueError("X and Y row counts differ")
    nbytes = int(lib.gpk_gpr_lml_workspace_bytes(n, d, P))
    if ws is None or ws.numel() * 8 < nbytes:
        ws = _ws(nbytes)
    out = torch.empty(1, dtype=torch.float64, device=X.device)
    info = torch.zeros(1, dtype=torch.int32, device=X.device)
    ls, ard = _ls_host(lengthscales, d)
    nv, nv_rows, _keep = _noise_args(noise_variance, n)
    rc = lib.gpk_gpr_lml(_stream(), KERNEL_FAMILIES[family], X.data_ptr(), n, d, _rowmajor(X, "X"),
                         Y.data_ptr(), P, _rowmajor(Y, "Y"), ls, ard, float(variance),
                         nv, nv_rows, float(mean_const), out.data_ptr(), info.data_ptr(),
                         ws.data_ptr(), ws.numel() * 8)
    _lib.check(rc, "gpk_gpr_lml")
    return out, info



class HostMailbox:
    """Scalars of a step delivered into pinned, device-mapped host memory by a kernel store (gpk_publish_host) instead of
    device-to-host copies + a stream synchronise: `post(src, info)` enqueues one tiny kernel behind the step on the current
    stream, `wait()` spins on the sequence word that kernel writes last and returns (values, info).  The reference reads
    the same scalar with `.numpy()` on the ELBO tensor (svgp.py:181 -> optimizers / monitoring); here that read costs a few
    microseconds instead of 90 - 150 us per step (two blit copies + hipStreamSynchronize, profiles/r03_step_timeline.txt).
    Falls back to nothing: a wait that times out raises."""

    def __init__(self, n: int = 2):
        if not 1 <= n <= 16:
            raise ValueError("HostMailbox holds 1..16 doubles")
        device()  # raises without a HIP device
        self.n = int(n)
        self._buf = torch.zeros(n + 1, dtype=torch.float64).pin_memory()
        self._vals = self._buf.numpy()[:n]
        self._tail = self._buf.numpy()[n:].view(np.int32)  # [info, seq]
        self._seq = 0

    def post(self, src: torch.Tensor, info: Optional[torch.Tensor] = None) -> int:
        lib = _lib.load()
        _chk(src, "src")
        if src.numel() < self.n or not src.is_contiguous():
            raise ValueError("src must be a contiguous tensor with at least n elements")
        if info is not None and (not info.is_cuda or info.dtype != torch.int32):
            raise _lib.GpkError("info must be an int32 tensor on the HIP device")
        self._seq = (self._seq % 0x7FFFFFF0) + 1
        rc = lib.gpk_publish_host(_stream(), src.data_ptr(), self.n, info.data_ptr() if info is not None else None,
                                  self._buf.data_ptr(), self._seq)
        _lib.check(rc, "gpk_publish_host")
        return self._seq

    def wait(self, timeout_s: float = 30.0):
        import time as _time
        tail, seq = self._tail, self._seq
        spins, t0 = 0, None
        while int(tail[1]) != seq:
            spins += 1
            if spins & 0x3FFF == 0:
                now = _time.perf_counter()
                t0 = t0 or now
                if now - t0 > timeout_s:
                    raise _lib.GpkError("HostMailbox.wait timed out (the publishing kernel never ran)")
        return self._vals.copy(), int(tail[0])


def svgp_elbo_sep_workspace(m: int, rows: int, d: int, P: int) -> torch.Tensor:
    lib = _lib.load()
    return _ws(int(lib.gpk_svgp_elbo_sep_workspace_bytes(m, rows, d, P)))


def svgp_elbo_shard_sep(Z: torch.Tensor, Xb: torch.Tensor, Yb: torch.Tensor, q_mu: torch.Tensor, q_sqrt: torch.Tensor, *,
                        variances, lengthscales, families, noise_variance, jitter: float, mean_const: float = 0.0,
                        ws: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                        info: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Whitened shard with one kernel PER latent (SeparateIndependent): Z [m, d] (shared) or [P, m, d]; variances [P],
    lengthscales [P] (isotropic) or [P, d], families [P] names; q_sqrt [P, m, m].  out[0] = sum_b var_exp_b, out[1] = KL;
    info [P].  Returns (out, info)."""
    lib = _lib.load()
    for name, t in (("Xb", Xb), ("Yb", Yb), ("q_mu", q_mu)):
        _chk(t, name, 2)
    _chk(Z, "Z")
    _chk(q_sqrt, "q_sqrt", 3)
    P = q_mu.shape[1]
    shared = Z.dim() == 2
    m, d = Z.shape[-2], Z.shape[-1]
    rows = Xb.shape[0]
    if (not shared and (Z.dim() != 3 or Z.shape[0] != P)) or Xb.shape[1] != d or Yb.shape[0] != rows or Yb.shape[1] != P \
            or q_mu.shape[0] != m or tuple(q_sqrt.shape) != (P, m, m) or len(families) != P:
        raise ValueError("inconsistent shapes")
    if not (Z.is_contiguous() and q_mu.is_contiguous() and q_sqrt.is_contiguous()):
        raise ValueError("Z / q_mu / q_sqrt must be contiguous")
    ls = np.asarray(lengthscales, dtype=np.float64)
    var = np.asarray(variances, dtype=np.float64).reshape(-1)
    if var.size != P or ls.shape[0] != P or (ls.ndim == 2 and ls.shape[1] != d) or ls.ndim > 2:
        raise ValueError("variances / lengthscales: one entry (row) per latent")
    ard = int(ls.ndim == 2)
    nbytes = int(lib.gpk_svgp_elbo_sep_workspace_bytes(m, rows, d, P))
    if ws is None or ws.numel() * 8 < nbytes:
        ws = _ws(nbytes)
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=Xb.device)
    if info is None:
        info = torch.zeros(P, dtype=torch.int32, device=Xb.device)
    fam = (_lib.C.c_int * P)(*[KERNEL_FAMILIES[f] for f in families])
    nv, nv_rows, _keep = _noise_args(noise_variance, rows)
    rc = lib.gpk_svgp_elbo_shard_sep(_stream(), fam, Z.data_ptr(), m, d, 0 if shared else m * d, Xb.data_ptr(), Yb.data_ptr(), rows,
                                     _rowmajor(Xb, "Xb"), _rowmajor(Yb, "Yb"), d, P, _lib.host_doubles(ls.reshape(-1).tolist()), ard,
                                     _lib.host_doubles(var.tolist()), nv, nv_rows, float(jitter), float(mean_const),
                                     q_mu.data_ptr(), q_sqrt.data_ptr(), out.data_ptr(), info.data_ptr(), ws.data_ptr(),
                                     ws.numel() * 8)
    _lib.check(rc, "gpk_svgp_elbo_shard_sep")
    return out, info


def svgp_elbo_workspace(m: int, rows: int, d: int, P: int, q_diag: bool, whiten: bool = True) -> torch.Tensor:
    lib = _lib.load()
    return _ws(int(lib.gpk_svgp_elbo_workspace_bytes(m, rows, d, P, int(q_diag), int(bool(whiten)))))


def svgp_elbo_shard(Z: torch.Tensor, Xb: torch.Tensor, Yb: torch.Tensor, q_mu: torch.Tensor,
                    q_sqrt: torch.Tensor, *, variance: float, lengthscales, noise_variance,
                    jitter: float, mean_const: float = 0.0, family: str = "SquaredExponential",
                    ws: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                    info: Optional[torch.Tensor] = None, whiten: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """One shard of SVGP.elbo: out[0] = sum_b var_exp_b over this shard, out[1] = KL.  Returns (out, info).
    whiten=False: KL against N(0, Kuu) and the un-whitened conditional, on one factorisation (full or diagonal q_sqrt)."""
    lib = _lib.load()
    for name, t in (("Z", Z), ("Xb", Xb), ("Yb", Yb), ("q_mu", q_mu)):
        _chk(t, name, 2)
    _chk(q_sqrt, "q_sqrt")
    m, d = Z.shape
    rows = Xb.shape[0]
    P = q_mu.shape[1]
    q_diag = q_sqrt.dim() == 2
    if Xb.shape[1] != d or Yb.shape[0] != rows or Yb.shape[1] != P or q_mu.shape[0] != m:
        raise ValueError("inconsistent shapes")
    if not (q_mu.is_contiguous() and q_sqrt.is_contiguous()):
        raise ValueError("q_mu / q_sqrt must be contiguous")
    nbytes = int(lib.gpk_svgp_elbo_workspace_bytes(m, rows, d, P, int(q_diag), int(bool(whiten))))
    if ws is None or ws.numel() * 8 < nbytes:
        ws = _ws(nbytes)
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=Z.device)
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=Z.device)
    ls, ard = _ls_host(lengthscales, d)
    nv, nv_rows, _keep = _noise_args(noise_variance, rows)
    rc = lib.gpk_svgp_elbo_shard(_stream(), KERNEL_FAMILIES[family], Z.data_ptr(), m, _rowmajor(Z, "Z"),
                                 Xb.data_ptr(), Yb.data_ptr(), rows, _rowmajor(Xb, "Xb"),
                                 _rowmajor(Yb, "Yb"), d, P, ls, ard, float(variance),
                                 nv, nv_rows, float(jitter), float(mean_const),
                                 q_mu.data_ptr(), q_sqrt.data_ptr(), int(q_diag), int(bool(whiten)), out.data_ptr(),
                                 info.data_ptr(), ws.data_ptr(), ws.numel() * 8)
    _lib.check(rc, "gpk_svgp_elbo_shard")
    return out, info
