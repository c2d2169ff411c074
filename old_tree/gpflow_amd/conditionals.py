"""Dense conditionals (gpflow/conditionals/util.py:37-169, 222-239, 566-629; conditionals.py:39-156).

Row-major convention on the device: the reference's A = Lm^-1 Kmn [M,N] is held transposed,
At = Kfu Lm^-T [N,M], so that every triangular solve is the right-side form the fused trapezoidal
Cholesky produces and every reduction over M is a contiguous row reduction.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import config, ops


@dataclass
class Factor:
    """Lower Cholesky factor + the inverses of its 128x128 diagonal blocks (what gpk_trsm needs)."""
    L: torch.Tensor
    invd: torch.Tensor
    _LT: Optional[torch.Tensor] = None
    _invdT: Optional[torch.Tensor] = None

    def transposed(self):
        if self._LT is None:
            self._LT, self._invdT = ops.transpose_factor(self.L, self.invd)
        return self._LT, self._invdT


def factor_with_rows(Kmm: torch.Tensor, rows: Optional[torch.Tensor], *, check: bool = True):
    """Cholesky of Kmm with `rows` [N,M] riding along: returns (Factor, rows Lm^-T).  One fused
    trapezoidal factorisation (tf.linalg.cholesky util.py:67 + triangular_solve util.py:125)."""
    M = Kmm.shape[0]
    n_extra = 0 if rows is None else rows.shape[0]
    T = torch.empty((M + n_extra, M), dtype=torch.float64, device=Kmm.device)
    T[:M].copy_(Kmm)
    if n_extra:
        T[M:].copy_(rows)
    invd, info = ops.potrf_(T, M, zero_upper=True)
    if check:
        ops.check_info(info)
    return Factor(T[:M], invd), (T[M:] if n_extra else None)


def _as_rows(Kmn: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, ...]]:
    """Kmn [M, batch..., N] -> (Kfu rows [prod(batch)*N, M] row-major, batch shape + (N,))
    (the reference moves M next to N with a transpose, util.py:108-119)."""
    M = Kmn.shape[0]
    lead = tuple(Kmn.shape[1:])
    if Kmn.dim() == 2 and Kmn.stride(0) == 1 and Kmn.stride(1) >= M:
        return Kmn.t(), lead  # already a transposed view of a row-major [N,M] buffer
    flat = Kmn.reshape(M, -1)
    return ops.transpose(flat.contiguous()), lead


def conditional_tail(At: torch.Tensor, fac: Factor, Knn: torch.Tensor, f: torch.Tensor, *,
                     full_cov: bool, q_sqrt: Optional[torch.Tensor], white: bool,
                     Linv_f: Optional[torch.Tensor] = None):
    """Everything of base_conditional_with_lm after A = Lm^-1 Kmn (util.py:128-167).
    At [N,M] (overwritten when not white); Knn [N] or [N,N]; f [M,R]; returns fmean [N,R] and
    fvar [N,R] / [R,N,N].
    Linv_f [M,R] = Lm^-1 f, if the caller has it (it rides through the factorisation as R extra rows): with no q_sqrt
    the un-whitened mean  (Lm^-T A)^T f = A^T (Lm^-1 f)  then needs no second triangular solve of the N columns of A
    (util.py:139 costs M^2 N flops; this form costs M N) -- the equivalent alpha-form SURVEY 8d names for GPR predict."""
    N, M = At.shape
    R = f.shape[1]
    f = f.contiguous()
    if not white and q_sqrt is None and Linv_f is not None:
        white, f = True, Linv_f.contiguous()
    diag_w = q_sqrt.contiguous() if (q_sqrt is not None and q_sqrt.dim() == 2 and not full_cov) else None
    one_pass = white and not full_cov      # A is not replaced: sum_k A^2, A^T f and the diagonal-q term share ONE read of A
    if full_cov:
        fvar0 = Knn - ops.gemm_nt(At, At)  # Knn - A^T A           (util.py:129)
    elif one_pass:
        s0, fmean, wsq = ops.row_stats(At, V=f, W=diag_w)
        fvar0 = Knn - s0
    else:
        s0, _, _ = ops.row_stats(At)  # sum_k A^2              (util.py:133)
        fvar0 = Knn - s0
    if not white:
        LT, invdT = fac.transposed()
        ops.trsm_(At, LT, invdT, trans=1)  # A <- Lm^-T A           (util.py:139)
    if not one_pass:
        # A^T f (util.py:144) and, for a diagonal q_sqrt, sum_k (A q_sqrt)^2 (util.py:149,164) in one pass
        _, fmean, wsq = ops.row_stats(At, V=f, W=diag_w, want_sumsq=False)
    if q_sqrt is None:
        if full_cov:
            fvar = fvar0[None].expand(R, N, N).contiguous()
        else:
            fvar = fvar0[:, None].expand(N, R).contiguous()
        return fmean, fvar
    if q_sqrt.dim() == 2:
        if full_cov:
            covs = []
            for r in range(R):
                W = (At * q_sqrt[:, r][None, :]).contiguous()  # LTA^T for a diagonal q_sqrt
                covs.append(fvar0 + ops.gemm_nt(W, W))
            return fmean, torch.stack(covs)
        return fmean, (fvar0[None, :] + wsq).t().contiguous()
    if q_sqrt.dim() != 3:
        raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.dim()))
    LqT = ops.transpose(q_sqrt.contiguous(), mode=1)  # band_part(q_sqrt,-1,0)^T   (util.py:151)
    if full_cov:
        covs = []
        for r in range(R):
            W = ops.gemm_nt(At, LqT[r], b_tri=1)  # (L^T A)^T            (util.py:157)
            covs.append(fvar0 + ops.gemm_nt(W, W))  # + LTA^T LTA        (util.py:162)
        return fmean, torch.stack(covs)
    ssq = ops.project(At.contiguous() if At.stride(1) != 1 else At, LqT)  # sum LTA^2 (util.py:164)
    return fmean, (fvar0[None, :] + ssq).t().contiguous()


def tail_over_batches(At: torch.Tensor, lead: Tuple[int, ...], knn_of, tail, full_cov: bool):
    """Leading batch dims of Xnew / Kmn (util.py:108-131) were flattened into the rows of At [prod(lead), M],
    lead = batch... + (T,).  Marginal variances: one call on all rows, reshaped to [batch..., T, R].  full_cov: the
    reference broadcasts A^T A over the batch dims and returns [batch..., R, T, T]; here the tail runs once per batch
    element on its T rows (a contiguous row block of At).  knn_of(b) -> Knn of batch element b (None: all rows,
    marginal); tail(At_rows, Knn) -> (fmean [rows, R], fvar)."""
    if len(lead) == 1:
        return tail(At, knn_of(None))
    if not full_cov:
        fmean, fvar = tail(At, knn_of(None))
        return fmean.reshape(*lead, -1), fvar.reshape(*lead, -1)
    T = lead[-1]
    nb = At.shape[0] // T
    mus, vs = [], []
    for b in range(nb):
        mu, var = tail(At[b * T:(b + 1) * T], knn_of(b))
        mus.append(mu)
        vs.append(var)
    fmean = torch.stack(mus).reshape(*lead, -1)
    fvar = torch.stack(vs)  # [nb, R, T, T]
    return fmean, fvar.reshape(*lead[:-1], *fvar.shape[1:])


def _knn_blocks(Knn: torch.Tensor, lead, full_cov: bool):
    """Knn as handed to base_conditional: [batch..., N] or [batch..., N, N] -> accessor per batch element."""
    if full_cov and len(lead) > 1:
        blocks = Knn.reshape(-1, lead[-1], lead[-1])
        return lambda b: blocks[b]
    flat = Knn if full_cov else Knn.reshape(-1)
    return lambda b: flat


def base_conditional(Kmn, Kmm, Knn, f, *, full_cov: bool = False, q_sqrt=None, white: bool = False):
    """gpflow/conditionals/util.py:37-70.  Kmn [M, batch..., N], Kmm [M,M], Knn [batch..., N] or
    [N,N], f [M,R], q_sqrt [M,R] | [R,M,M] | None."""
    Kmn, Kmm, Knn, f = (ops.to_device(t) for t in (Kmn, Kmm, Knn, f))
    q_sqrt = ops.to_device(q_sqrt) if q_sqrt is not None else None
    rows, lead = _as_rows(Kmn)
    fac, At = factor_with_rows(Kmm, rows)
    return tail_over_batches(At, lead, _knn_blocks(Knn, lead, full_cov),
                             lambda A, K: conditional_tail(A, fac, K, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white),
                             full_cov)


def base_conditional_with_lm(Kmn, Lm, Knn, f, *, full_cov: bool = False, q_sqrt=None,
                             white: bool = False, _factor: Optional[Factor] = None):
    """gpflow/conditionals/util.py:84-169 (Lm precomputed)."""
    Kmn, Lm, Knn, f = (ops.to_device(t) for t in (Kmn, Lm, Knn, f))
    q_sqrt = ops.to_device(q_sqrt) if q_sqrt is not None else None
    fac = _factor if _factor is not None else Factor(Lm.contiguous(), ops.trtri_blocks(Lm.contiguous()))
    rows, lead = _as_rows(Kmn)
    At = rows.contiguous().clone() if rows.data_ptr() == Kmn.data_ptr() else rows.contiguous()
    ops.trsm_(At, fac.L, fac.invd, trans=0)  # A = Lm^-1 Kmn (util.py:125)
    return tail_over_batches(At, lead, _knn_blocks(Knn, lead, full_cov),
                             lambda A, K: conditional_tail(A, fac, K, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white),
                             full_cov)


def expand_independent_outputs(fvar: torch.Tensor, full_cov: bool, full_output_cov: bool) -> torch.Tensor:
    """gpflow/conditionals/util.py:222-239 (shape glue)."""
    if full_cov and full_output_cov:
        fvar = torch.diag_embed(fvar.permute(*range(fvar.dim() - 3), -2, -1, -3))  # [..., N, N, P, P]
        fvar = fvar.transpose(-3, -2)  # [..., N, P, N, P]
    if not full_cov and full_output_cov:
        fvar = torch.diag_embed(fvar)  # [..., N, P, P]
    return fvar


def separate_independent_conditional_implementation(Kmns, Kmms, Knns, f, *, full_cov: bool = False,
                                                    q_sqrt=None, white: bool = False):
    """gpflow/conditionals/util.py:566-629: P independent GPs.  The reference loops with tf.map_fn;
    here all P Cholesky factorisations + solves run as ONE batched trapezoidal factorisation.
    Kmns [P,M,N], Kmms [P,M,M], Knns [P,N] or [P,N,N], f [M,P], q_sqrt [M,P] | [P,M,M] | None."""
    Kmns, Kmms, Knns, f = (ops.to_device(t) for t in (Kmns, Kmms, Knns, f))
    q_sqrt = ops.to_device(q_sqrt) if q_sqrt is not None else None
    P, M, N = Kmns.shape
    T = torch.empty((P, M + N, M), dtype=torch.float64, device=Kmms.device)
    T[:, :M].copy_(Kmms)
    T[:, M:].copy_(Kmns.transpose(1, 2))
    return separate_independent_trapezoid_tail(T, M, Knns, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white)


def separate_independent_trapezoid_tail(T: torch.Tensor, M: int, Knns, f, *, full_cov: bool, q_sqrt, white: bool):
    """The same from the batched trapezoid T [P, M + N, M] = [Kmm_p ; Kfu_p] (consumed): callers that can BUILD the
    covariances straight into T (the posteriors) skip the [P,M,M] / [P,M,N] intermediates and their copies."""
    P = T.shape[0]
    invd, info = ops.potrf_(T, M, zero_upper=True)
    ops.check_info(info)
    invd = invd.reshape(P, -1)
    if not full_cov and q_sqrt is not None and q_sqrt.dim() == 3:
        # marginal variances with full q_sqrt (the SVGP ELBO / predict_f of BASELINE config C5 with separate kernels): the
        # P projections onto q_sqrt_p are ONE launch over the batched trapezoid (gpk_project_batched)
        s0s, mus = [], []
        for p in range(P):
            At = T[p, M:]
            fp = f[:, p:p + 1].contiguous()
            if white:
                s0, mu, _ = ops.row_stats(At, V=fp)
            else:
                s0, _, _ = ops.row_stats(At)
                LT, invdT = Factor(T[p, :M], invd[p]).transposed()
                ops.trsm_(At, LT, invdT, trans=1)
                _, mu, _ = ops.row_stats(At, V=fp, want_sumsq=False)
            s0s.append(s0)
            mus.append(mu[:, 0])
        ssq = ops.project(T[:, M:], ops.transpose(q_sqrt.contiguous(), mode=1))           # [P, rows]
        fvar = torch.stack([Knns[p] - s0s[p] for p in range(P)]) + ssq
        return torch.stack(mus, dim=-1), fvar.t().contiguous()
    mus, vs = [], []
    for p in range(P):
        fac = Factor(T[p, :M], invd[p])
        qs = None
        if q_sqrt is not None:
            qs = q_sqrt[:, p:p + 1] if q_sqrt.dim() == 2 else q_sqrt[p:p + 1]
        mu, var = conditional_tail(T[p, M:], fac, Knns[p], f[:, p:p + 1].contiguous(), full_cov=full_cov,
                                   q_sqrt=qs, white=white)
        mus.append(mu[:, 0])
        vs.append(var[0] if full_cov else var[:, 0])
    fmu = torch.stack(mus, dim=-1)
    fvar = torch.stack(vs, dim=0) if full_cov else torch.stack(vs, dim=-1)
    return fmu, fvar


def conditional(Xnew, inducing_variable, kernel, f, *, full_cov: bool = False,
                full_output_cov: bool = False, q_sqrt=None, white: bool = False):
    """gpflow/conditionals/conditionals.py:39-87 (inducing variables: routes through the posterior's
    fused path, so conditional() and fused_predict_f are the same code) and :101-156 (X data)."""
    from .inducing_variables import InducingVariables
    from . import posteriors

    if isinstance(inducing_variable, InducingVariables):
        posterior = posteriors.create_posterior(kernel, inducing_variable, f, q_sqrt, white,
                                                mean_function=None, precompute_cache=None)
        return posterior.fused_predict_f(Xnew, full_cov=full_cov, full_output_cov=full_output_cov)
    # dense: condition on function values f at the data X (conditionals.py:143-156)
    X = ops.to_device(inducing_variable)
    Xnew = ops.to_device(Xnew)
    Xs, Xn = kernel.slice(X, Xnew)
    Kmm = kernel.K_into(Xs, None, None, diag_add=config.default_jitter())
    Kfu = kernel.K_into(Xn, Xs, None)
    Knn = kernel(Xnew, full_cov=full_cov)
    mean, var = base_conditional(Kfu.t(), Kmm, Knn, ops.to_device(f), full_cov=full_cov, q_sqrt=q_sqrt,
                                 white=white)
    return mean, var
