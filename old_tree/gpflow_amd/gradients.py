"""Reverse pass of the SVGP hot path (what turns "ELBO steps/s" into "training steps/s").

The reference gets these gradients from TensorFlow autodiff over `SVGP.elbo` (`optimizers/scipy.py:322-331`,
`models/training_mixins.py:59-78`, the Adam loop of `gps_for_big_data.pct.py:207-228`).  Here the adjoint of each
stage of the whitened ELBO is written out and executed with the same device primitives as the forward pass --
the fp64 MFMA GEMM (plain / triangular-K / lower-only), the trapezoidal Cholesky with its fused solve (which also
delivers Lm^-T from appended identity rows, so no triangular solve is left in the backward), the covariance builder (and its `G .* K` variant) -- so the backward is, like the
forward, a short list of big GEMM-shaped launches:

    forward   Kuu, Kfu -> Lm, At = Kfu Lm^-T -> fmean = At q_mu, s0 = rowsum(At^2), W_p = At Lq_p, ssq = rowsum(W_p^2)
              F = scale * sum var_exp(fmean, sigma^2 - s0 + ssq) - KL[q || N(0, I)]
    backward  At_bar = r q_mu^T - 2 c P At + 2 c sum_p W_p Lq_p^T              r = dF/dfmean, c = dF/dfvar
              q_mu_bar = At^T r - q_mu,   Lq_bar_p = 2 c tril(At^T W_p) - tril(Lq_p) + diag(1 / Lq_p)
              Kfu_bar = At_bar Lm^-1,     Lm_bar = -tril(Kfu_bar^T At)
              Kuu_bar = sym( Lm^-T Phi(Lm^T Lm_bar) Lm^-1 )                    (Cholesky adjoint; Phi = tril, half diagonal)
              kernel parameters and Z from  G = Kbar .* K  contracted with [1, x, x^2]   (SquaredExponential)

Small O(M^2) / O(B P) elementwise steps (masks, the residual r, the final [M, 2D+1] contractions) are torch
device ops -- glue between the kernels, never a fallback: every function here needs the HIP library.

All arrays are fp64 device tensors; gradients are returned w.r.t. the CONSTRAINED quantities (variance,
lengthscales, noise variance, Z, q_mu, q_sqrt); `SVGP.elbo_and_grad` chains them through the parameter transforms.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import collections
import os

import numpy as np
import torch

from . import ops

LOG2PI = float(np.log(2.0 * np.pi))

# The reverse pass has two independent branches after Kfu_bar: {q_mu_bar, Lq_bar} (two long-K products, 544 workgroups
# each) and {Lm_bar -> Cholesky adjoint -> Kuu adjoint} (M^3 products of 256 tiles: one workgroup per CU, time of the
# longest tile).  On a device they run on two streams so that the under-filled launches share the chip.
OVERLAP_BRANCHES = True
# the side branch's chip-filling GEMMs start only when the main branch's HBM-bound preparation is enqueued (svgp_elbo_and_grad)
# K chunks of the triangular x triangular products of the Cholesky adjoint (1 = unsplit)
TRI_PRODUCT_CHUNKS = int(os.environ.get("GPFLOW_AMD_TRI_PRODUCT_CHUNKS", "4"))
TRI_PRODUCT_MIN_N = 1024   # below this the unsplit launches are short anyway
GATE_SIDE_BRANCH = os.environ.get("GPFLOW_AMD_GATE_SIDE_BRANCH", "1") != "0"
_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(dev: torch.device):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _side_streams:
        # (a LOW-priority side stream -- hipStreamCreateWithPriority, wrapped as an ExternalStream -- was tried in round 6 so that the main
        #  branch's short kernels would win the freed slots: training step 6.02 -> 6.9 ms, profiles/r06_ab_train_side_branch.log)
        _side_streams[idx] = torch.cuda.Stream(device=dev)
    return _side_streams[idx]


def splitk_gemm_nt(A: torch.Tensor, Bt: torch.Tensor, *, c_lower: bool = False, target_wgs: int = 1100,
                   alpha: float = 1.0, diag_scale: float = 1.0) -> torch.Tensor:
    """alpha A Bt^T for a LONG inner dimension and few output tiles (At^T r, At^T W, G [1, x, x^2]): the K range is cut
    into chunks that run as the batch dimension of one launch (strided views, no copies) and the partial products are
    summed by ops.combine_parts in a fixed order -- without the split a [2048, 2048] lower-only output is 136 workgroups
    each walking K = 8192 (33 TFLOP/s), and an [M, 17] output is 16 workgroups.  target_wgs ~ two full rounds of the 512
    resident workgroups (A/B on the training step: 272 workgroups 8.08 ms, 544 7.9, 1088 7.65, 2176 7.78).
    c_lower=True returns tril(.) (exact zeros above the diagonal) with the diagonal multiplied by diag_scale."""
    m, k = A.shape
    n = Bt.shape[0]
    tm, tn = -(-m // 128), -(-n // 128)
    tiles = tm * tn
    if c_lower:   # tiles on or below the diagonal
        tiles = sum(min(tn, i + 1) for i in range(tm))
    chunks = 1
    while chunks * 2 * tiles <= target_wgs and k % (chunks * 2) == 0 and (k // (chunks * 2)) % 16 == 0 \
            and k // (chunks * 2) >= 256:
        chunks *= 2
    if chunks == 1:
        R = ops.gemm_nt(A, Bt, c_lower=c_lower, alpha=alpha, zero_skipped=False)   # (combine_parts(lower) never reads above the diagonal)
        return ops.combine_parts(R, lower=True, diag_scale=diag_scale) if c_lower else R
    kc = k // chunks
    A3 = torch.as_strided(A, (chunks, m, kc), (kc, A.stride(0), 1), A.storage_offset())
    B3 = torch.as_strided(Bt, (chunks, n, kc), (kc, Bt.stride(0), 1), Bt.storage_offset())
    return ops.combine_parts(ops.gemm_nt(A3, B3, c_lower=c_lower, zero_skipped=False), alpha=alpha, lower=c_lower, diag_scale=diag_scale)


def _tril(x: torch.Tensor) -> torch.Tensor:
    """band_part(x, -1, 0) of [M, M] or [P, M, M] as a fresh tensor (one 16-byte-access pass per matrix)."""
    if x.dim() == 2:
        return ops.combine_parts(x, lower=True)
    out = torch.empty_like(x)
    for p in range(x.shape[0]):
        ops.combine_parts(x[p], lower=True, out=out[p])
    return out


def _phi_(T: torch.Tensor) -> torch.Tensor:
    """Phi(T): lower triangle with the diagonal halved (a fresh tensor)."""
    return ops.combine_parts(T, lower=True, diag_scale=0.5)


def cholesky_adjoint(LT: torch.Tensor, LinvT: torch.Tensor, Lbar: torch.Tensor, before_products=None) -> torch.Tensor:
    """Symmetric K_bar with  <K_bar, dK> = <L_bar, dL>  for K = L L^T:  K_bar = sym(L^-T Phi(L^T L_bar) L^-1), with
    LT = L^T, LinvT = L^-T (both upper, zero below the diagonal) and L_bar lower.  Three triangular-K GEMMs; the
    explicit inverse comes for free from the factorisation (identity rows appended to the trapezoid).
    before_products: called once the HBM-bound preparation is enqueued (see svgp_elbo_and_grad: the side branch's big GEMM is gated on it)."""
    # every B operand below is the transpose of a lower-triangular matrix: B[j, kk] = 0 for kk < j  -> b_tri = 1
    # (a_tri: A is triangular as well -- upper L^T / L^-T, lower Phi -- so a tile's K range is the intersection of both)
    LbarT = ops.transpose(Lbar)
    if before_products is not None:
        before_products()
    n = LT.shape[0]
    chunks = TRI_PRODUCT_CHUNKS if (TRI_PRODUCT_CHUNKS > 1 and n >= TRI_PRODUCT_MIN_N and n % (16 * TRI_PRODUCT_CHUNKS) == 0) else 1
    if chunks == 1:
        T1 = ops.gemm_nt(LT, LbarT, b_tri=1, a_tri=1)                        # L^T L_bar
        Y = ops.gemm_nt(_phi_(T1), LinvT, b_tri=1, a_tri=2)                  # Phi L^-1        (lower)
        S = ops.gemm_nt(LinvT, ops.transpose(Y, mode=1), b_tri=1, a_tri=1)   # L^-T (Phi L^-1)
        return ops.symmetrize_(S)
    # Each of the three products is 256 output tiles whose K ranges differ by a factor of sixteen -- the launch lasts as long as the tile
    # that walks all of K (245 us at M = 2048 for 5.7 GFLOP).  As K chunks in the batch dimension of ONE launch (strided views; the
    # triangular hints refer to the unsplit column index, ops.gemm_nt k_split) the non-empty tiles fill the chip about once, and the sum of
    # the partial products is the pass that applied Phi / tril anyway.
    kc = n // chunks

    def split(X):
        return torch.as_strided(X, (chunks, X.shape[0], kc), (kc, X.stride(0), 1), X.storage_offset())

    T1 = ops.combine_parts(ops.gemm_nt(split(LT), split(LbarT), b_tri=1, a_tri=1, k_split=True), lower=True, diag_scale=0.5)   # Phi(L^T L_bar)
    Y = ops.combine_parts(ops.gemm_nt(split(T1), split(LinvT), b_tri=1, a_tri=2, k_split=True), lower=True)                   # Phi L^-1 (lower)
    S = ops.combine_parts(ops.gemm_nt(split(LinvT), split(ops.transpose(Y)), b_tri=1, a_tri=1, k_split=True))               # L^-T (Phi L^-1)
    return ops.symmetrize_(S)


_ls_cache: "collections.OrderedDict" = collections.OrderedDict()


def ls_device(lengthscales, D: int, device) -> torch.Tensor:
    """Lengthscales as a [D] device tensor.  A host array -> device copy from pageable memory BLOCKS the host until the stream
    has drained; in the tail of a reverse pass that serialised ~80 short launches behind an idle GPU (1 ms of a 6.5-ms training
    step, profiles/r04_train_timeline_before.txt).  The tensors are therefore made once per set of values -- the gradient entry
    points ask for them BEFORE they enqueue their first kernel -- and found here by the adjoints."""
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (D,)))
    key = (a.tobytes(), str(device))
    t = _ls_cache.get(key)
    if t is None:
        t = torch.as_tensor(a.copy(), device=device)
        _ls_cache[key] = t
        while len(_ls_cache) > 32:
            _ls_cache.popitem(last=False)
    return t


def stationary_kernel_adjoint(A: torch.Tensor, Bm: torch.Tensor, Kbar: torch.Tensor, *, variance: float, lengthscales,
                              symmetric: bool, family: str = "SquaredExponential", packed_into=None, dvar_add: float = 0.0,
                              return_packed: bool = False):
    """Adjoint of K = variance * f(r(A / ls, Bm / ls)) [n1, n2] given Kbar [n1, n2], f one of the stationary families of
    `ops.KERNEL_FAMILIES` (stationaries.py:209-210 SquaredExponential, :254-313 Matern12 / 32 / 52).

    Returns (d/dvariance, d/dlengthscales [D], A_bar [n1, D]); with symmetric=True (Bm is A, Kbar symmetric) A_bar
    collects both arguments.  B_bar of the non-symmetric case is not needed on this path (B = the minibatch).

    With r2 the scaled squared distance, dF/dtheta = sum_ij Kbar_ij dK_ij/dr2 dr2_ij/dtheta.  G = Kbar .* (-2 dK/dr2) is
    one elementwise pass that recomputes r2 from the inputs (`gpk_kernel_matrix_combine` op 3); for the
    SquaredExponential -2 dK/dr2 = K, so G also carries d/dvariance = sum(Kbar .* K) / variance; the Matern families
    take that sum from a second pass (op 1)."""
    n1, D = A.shape
    ls = ls_device(lengthscales, D, A.device)
    if family == "SquaredExponential":
        G = ops.kernel_matrix_hadamard(A, Bm, Kbar, variance=variance, lengthscales=lengthscales)   # Kbar .* K
        sum_kbar_k = None
    else:
        G = ops.kernel_matrix_hadamard(A, Bm, Kbar, variance=variance, lengthscales=lengthscales, family=family)
        sum_kbar_k = G.sum()
        ops.kernel_matrix_combine(A, None if symmetric else Bm, Kbar, op="dr2", variance=variance,
                                  lengthscales=lengthscales, family=family, out=G)
    Vt = ops.moment_rows(Bm)                         # [1, B, B^2]^T
    R = splitk_gemm_nt(G, Vt)                        # [n1, 1 + 2D] = G [1, B, B^2]: row sums rs, G B, G B^2
    # one launch for what follows (it was a dozen elementwise / reduction launches on [n1, D] arrays):  T = G B - A rs;
    # both arguments (symmetric):  A_bar = 2 T / ls^2,  d/dls = (2 sum A^2 rs - 2 sum A GB) / ls^3 = -sum(A A_bar) / ls
    # else:  A_bar = T / ls^2,  d/dls = sum(GB2 - 2 A GB + A^2 rs) / ls^3 = sum(GB2 - A (GB + T)) / ls^3;  d/dvariance = sum(rs) / variance
    # (packed_into = (small [1 + D], A_bar) of an earlier adjoint of the SAME kernel: the results are added to it in the same launch;
    #  return_packed: (small, A_bar) instead of the three views)
    dvar, dls, Abar = ops.stationary_adjoint_tail(R, A if A.is_contiguous() else A.contiguous(), ls, variance=variance,
                                                  symmetric=symmetric, sum_kbar_k=sum_kbar_k, into=packed_into, dvar_add=dvar_add)
    if return_packed:
        return (packed_into[0] if packed_into is not None else dls._base), Abar
    return dvar, dls, Abar


se_kernel_adjoint = stationary_kernel_adjoint   # (the name of rounds 1-2)


class KernelSpec:
    """What the reverse pass needs to know about the covariance function: ONE stationary kernel, or a Sum / Product of
    stationary kernels -- flat, or nested (a Product of Sums, ...: `op` a tree, see __init__) -- over the same or different input columns (gpflow/kernels/base.py:216-220 `Sum`, :305-315 `Product`; the reference
    differentiates `tf.add_n` / `tf.multiply` of the member matrices with autodiff).  members: [(family, variance,
    lengthscales)], op: None | "add" | "mul".

      build    the combined matrix, members folded in place by `gpk_kernel_matrix_combine` (no second matrix);
      kdiag    K(x, x): sum or product of the variances (stationary members), dkdiag[i] = d kdiag / d variance_i;
      adjoint  member i sees  Kbar (Sum)  or  Kbar .* prod_{j != i} k_j  (Product: formed by the same in-place combine pass)
               and goes through `stationary_kernel_adjoint`; the input gradients of the members add up."""

    def __init__(self, members, op=None, cols=None):
        """members: [(family, variance, lengthscales)]; cols[i]: the input columns member i sees (its `active_dims` as an index
        list, kernels/base.py:90-109) or None for all of them.  Members that see different columns are built and differentiated on
        their own column slices; their input gradients are scattered back into the full columns and add up."""
        self.members = [(f, float(v), np.asarray(ls, dtype=np.float64)) for f, v, ls in members]
        # op: "add" | "mul" for a flat combination of all members, or a TREE for nested ones (a Product of Sums, ...):
        # (op, [children]) with a child either a member index or another such pair -- e.g. ("mul", [("add", [0, 1]), 2]) is
        # (k0 + k1) * k2 (kernels/base.py:223-329: a Combination holds kernels, which may be Combinations of the other kind)
        self.tree = None
        if isinstance(op, (tuple, list)):
            self.tree = self._check_tree(op, len(self.members))
            op = None
        self.op = op if len(self.members) > 1 else None
        if self.tree is None and len(self.members) > 1 and op not in ("add", "mul"):
            raise ValueError("a kernel combination needs op 'add' or 'mul'")
        cols = [None] * len(self.members) if cols is None else list(cols)
        if len(cols) != len(self.members):
            raise ValueError("one column selection per member")
        self.cols = [None if c is None else np.asarray(c, dtype=np.int64).reshape(-1) for c in cols]
        self._idx = {}

    @staticmethod
    def single(variance, lengthscales, family="SquaredExponential"):
        return KernelSpec([(family, variance, lengthscales)], None)

    @staticmethod
    def _check_tree(node, n):
        seen = []

        def walk(nd):
            if isinstance(nd, (int, np.integer)):
                seen.append(int(nd))
                return int(nd)
            op, ch = nd
            if op not in ("add", "mul") or len(ch) < 1:
                raise ValueError("a combination node is ('add' | 'mul', [children])")
            return (op, [walk(c) for c in ch])
        out = walk(node)
        if sorted(seen) != list(range(n)):
            raise ValueError("every member must appear exactly once in the combination tree")
        return out

    # ---- nested combinations: the same three operations, recursively over the tree -----------------------------------
    def _build_node(self, node, X1, X2, out):
        if isinstance(node, int):
            f, v, ls = self.members[node]
            return ops.kernel_matrix(self._sl(node, X1), self._sl(node, X2), variance=v, lengthscales=ls, family=f, diag_add=0.0,
                                     lower_only=False, out=out)
        op, ch = node
        out = self._build_node(ch[0], X1, X2, out)
        for c in ch[1:]:
            if isinstance(c, int):      # a stationary member folds in place (K recomputed in registers, no second matrix)
                f, v, ls = self.members[c]
                ops.kernel_matrix_combine(self._sl(c, X1), self._sl(c, X2), out, op=op, variance=v, lengthscales=ls, family=f,
                                          diag_add=0.0, out=out)
            else:                        # a sub-combination needs its own matrix once
                tmp = self._build_node(c, X1, X2, None)
                out.add_(tmp) if op == "add" else out.mul_(tmp)
        return out

    def _kd_node(self, node) -> float:
        if isinstance(node, int):
            return self.members[node][1]
        vals = [self._kd_node(c) for c in node[1]]
        return float(np.prod(vals)) if node[0] == "mul" else float(np.sum(vals))

    def _dkd_node(self, node) -> dict:
        if isinstance(node, int):
            return {node: 1.0}
        op, ch = node
        out = {}
        kds = [self._kd_node(c) for c in ch]
        for ci, c in enumerate(ch):
            fac = float(np.prod([k for j, k in enumerate(kds) if j != ci])) if op == "mul" else 1.0
            for i, d in self._dkd_node(c).items():
                out[i] = fac * d
        return out

    def _adjoint_node(self, node, A, Bm, Kbar, symmetric, acc):
        if isinstance(node, int):
            f, v, ls = self.members[node]
            Ai = self._sl(node, A)
            Bi = Ai if (symmetric and Bm is A) else self._sl(node, Bm)
            dv, dl, Ab = stationary_kernel_adjoint(Ai, Bi, Kbar, symmetric=symmetric, variance=v, lengthscales=ls, family=f)
            if np.ndim(ls) == 0 or np.size(ls) == 1:
                dl = dl.sum().reshape(1)
            acc[node] = (dv.reshape(1), dl, Ab)
            return
        op, ch = node
        for c in ch:
            Kb = Kbar
            if op == "mul":          # d/dK_c = Kbar .* prod of the OTHER children's matrices
                Kb = Kbar.clone()
                for o in ch:
                    if o is c:
                        continue
                    if isinstance(o, int):
                        fo, vo, lo = self.members[o]
                        ops.kernel_matrix_combine(self._sl(o, A), None if symmetric else self._sl(o, Bm), Kb, op="mul", variance=vo,
                                                  lengthscales=lo, family=fo, out=Kb)
                    else:
                        Kb.mul_(self._build_node(o, A, None if symmetric else Bm, None))
            self._adjoint_node(c, A, Bm, Kb, symmetric, acc)

    @property
    def n(self) -> int:
        return len(self.members)

    def _sl(self, i: int, X):
        """member i's view of the inputs: the columns of its active_dims as a contiguous copy (None stays None)"""
        if X is None or self.cols[i] is None:
            return X
        key = (i, str(X.device))
        if key not in self._idx:
            self._idx[key] = torch.as_tensor(self.cols[i], device=X.device)
        return X.index_select(1, self._idx[key]).contiguous()

    def build(self, X1, X2, out=None, diag_add: float = 0.0):
        if self.tree is not None:
            out = self._build_node(self.tree, X1, X2, out)
            if diag_add != 0.0 and X2 is None:
                out.diagonal().add_(float(diag_add))
            return out
        last = self.n - 1
        f, v, ls = self.members[0]
        out = ops.kernel_matrix(self._sl(0, X1), self._sl(0, X2), variance=v, lengthscales=ls, family=f,
                                diag_add=diag_add if last == 0 else 0.0, lower_only=False, out=out)
        for i, (f, v, ls) in enumerate(self.members[1:], start=1):
            ops.kernel_matrix_combine(self._sl(i, X1), self._sl(i, X2), out, op=self.op, variance=v, lengthscales=ls, family=f,
                                      diag_add=diag_add if i == last else 0.0, out=out)
        return out

    def kdiag(self) -> float:
        if self.tree is not None:
            return self._kd_node(self.tree)
        vs = [v for _, v, _ in self.members]
        return float(np.prod(vs)) if self.op == "mul" else float(np.sum(vs))

    def dkdiag(self):
        if self.tree is not None:
            d = self._dkd_node(self.tree)
            return [d[i] for i in range(self.n)]
        if self.op == "mul":
            kd = self.kdiag()
            return [kd / v for _, v, _ in self.members]
        return [1.0] * self.n

    def warm(self, device, D: int) -> "KernelSpec":
        """device copies of the members' lengthscales, made now (see `ls_device`)"""
        for i, (_, _, ls) in enumerate(self.members):
            ls_device(ls, D if self.cols[i] is None else len(self.cols[i]), device)
        return self

    def adjoint(self, A, Bm, Kbar, symmetric: bool):
        dvs, dls, Abar = [], [], None
        full = all(c is None for c in self.cols)
        if self.tree is not None:
            acc = {}
            self._adjoint_node(self.tree, A, Bm, Kbar, symmetric, acc)
            for i in range(self.n):
                dv, dl, Ab = acc[i]
                dvs.append(dv)
                dls.append(dl)
                if full:
                    Abar = Ab if Abar is None else Abar + Ab
                else:
                    if Abar is None:
                        Abar = torch.zeros_like(A)
                    if self.cols[i] is None:
                        Abar += Ab
                    else:
                        self._sl(i, A)    # (makes sure the index tensor of member i exists on this device)
                        Abar.index_add_(1, self._idx[(i, str(A.device))], Ab)
            return dvs, dls, Abar
        for i, (f, v, ls) in enumerate(self.members):
            Kb = Kbar
            if self.op == "mul":
                Kb = Kbar.clone()
                for j, (fj, vj, lj) in enumerate(self.members):
                    if j != i:
                        ops.kernel_matrix_combine(self._sl(j, A), None if symmetric else self._sl(j, Bm), Kb, op="mul", variance=vj,
                                                  lengthscales=lj, family=fj, out=Kb)
            Ai = self._sl(i, A)
            Bi = Ai if (symmetric and Bm is A) else self._sl(i, Bm)
            dv, dl, Ab = stationary_kernel_adjoint(Ai, Bi, Kb, symmetric=symmetric, variance=v, lengthscales=ls, family=f)
            if np.ndim(ls) == 0 or np.size(ls) == 1:
                dl = dl.sum().reshape(1)
            dvs.append(dv.reshape(1))
            dls.append(dl)
            if full:
                Abar = Ab if Abar is None else Abar + Ab
            else:   # scatter the member's input gradient back into the columns it read
                if Abar is None:
                    Abar = torch.zeros_like(A)
                if self.cols[i] is None:
                    Abar += Ab
                else:
                    Abar.index_add_(1, self._idx[(i, str(A.device))], Ab)
        return dvs, dls, Abar

    def pack(self, dvs, dls):
        """gradient entries: one member -> ("variance" [1], "lengthscales" [D or 1]) as before; several -> "variance" [n] and
        "lengthscales" a list of per-member tensors"""
        if self.n == 1:
            return dvs[0], dls[0]
        return torch.cat(dvs), list(dls)



def svgp_elbo_and_grad(Z: torch.Tensor, Xb: torch.Tensor, Yb: torch.Tensor, q_mu: torch.Tensor, q_sqrt: torch.Tensor,
                       *, variance: float = None, lengthscales=None, noise_variance: float, jitter: float, scale: float = 1.0,
                       mean_const: float = 0.0, kl_weight: float = 1.0, family: str = "SquaredExponential",
                       kernel_spec: "KernelSpec" = None
                       ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """F = scale * sum_b var_exp_b - kl_weight * KL for the whitened SVGP with a stationary kernel (or, `kernel_spec`, a Sum /
    Product of stationary kernels: then grads["variance"] is [n_members] and grads["lengthscales"] a list), a Gaussian likelihood
    and a full q_sqrt [P, M, M], and dF/d{variance, lengthscales, noise_variance, Z, q_mu, q_sqrt, mean_const}.

    Returns (F [1], grads, info).  `scale` = num_data / minibatch size (svgp.py:176-180).  For a row shard of a
    data-parallel step pass the GLOBAL scale and kl_weight = 1 / world_size: the SUM over ranks of F and of every
    gradient is then the full-batch value (one all-reduce of the packed gradient, distributed.all_reduce_grads)."""
    M, D = Z.shape
    B = Xb.shape[0]
    P = q_mu.shape[1]
    q_diag = q_sqrt.dim() == 2
    if tuple(q_sqrt.shape) not in ((P, M, M), (M, P)):
        raise ValueError("svgp_elbo_and_grad needs q_sqrt [P, M, M] or, for q_diag, [M, P]")
    dev = Z.device
    spec = kernel_spec if kernel_spec is not None else KernelSpec.single(variance, lengthscales, family)
    spec.warm(dev, D)   # (device copies of the lengthscales BEFORE the first kernel is enqueued: see ls_device)

    # ---------------------------------------------------------------- forward (intermediates kept)
    # trapezoid = [Kuu + jitter I ; Kfu ; I]: the factorisation returns Lm, At = Kfu Lm^-T and, from the identity
    # rows, Lm^-T itself -- the explicit inverse that turns every triangular solve of the backward into one GEMM
    T = torch.empty((M + B + M, M), dtype=torch.float64, device=dev)
    spec.build(Z, None, T[:M], diag_add=jitter)                                         # Kuu + jitter I
    spec.build(Xb, Z, T[M:M + B])                                                       # Kfu
    invd, info = ops.potrf_(T, M, zero_upper=True, identity_rows=True)                  # (the last M rows: I -> Lm^-T)
    L, At, LinvT = T[:M], T[M:M + B], T[M + B:]
    if q_diag:   # q_sqrt [M, P] holds standard deviations (svgp.py:90-148, conditionals/util.py:149,164)
        s0, fmean, ssq = ops.row_stats(At, V=q_mu, W=q_sqrt.contiguous())                # rowsum(At^2), At q_mu, sum_k At^2 q^2
        Lq = LqT = W = None
    else:
        Lq = _tril(q_sqrt)                                                                 # band_part(q_sqrt, -1, 0)
        LqT = ops.transpose(q_sqrt, mode=1)                                             # [P, M, M] = tril(q_sqrt)^T
        s0, fmean, _ = ops.row_stats(At, V=q_mu)                                        # rowsum(At^2), At q_mu
        W = ops.gemm_nt(At, LqT, b_tri=1)                                               # [P, B, M]: W_p = At Lq_p
        ssq = torch.stack([ops.row_stats(W[p])[0] for p in range(P)])                   # [P, B]
    ve, _ = ops.gaussian_varexp_sum(Yb, fmean, s0=s0, ssq=ssq, knn=[spec.kdiag()], noise_variance=noise_variance,
                                    mean_const=mean_const)
    kl = ops.gauss_kl_white(q_mu, q_sqrt)
    F = torch.mul(ve, scale).sub_(kl, alpha=kl_weight)

    # ---------------------------------------------------------------- backward
    # het: one noise variance per row (a heteroskedastic Gaussian likelihood, round 5): dF/dfvar is a per-row vector, applied as a
    # row scaling of the factors it multiplied as a scalar (elementwise glue on [B, M] arrays); the scalar path is unchanged
    het = torch.is_tensor(noise_variance) and noise_variance.dim() >= 1
    if het:
        nv = noise_variance.reshape(-1)
        cvec = (-0.5 * scale) / nv                                                      # dF/dfvar per row [B]
        c = None
        r = (scale / nv)[:, None] * (Yb - fmean - mean_const)                           # dF/dfmean [B, P]
    else:
        c = -0.5 * scale / noise_variance                                               # dF/dfvar (every b, p)
        r = torch.sub(Yb, fmean)                                                        # dF/dfmean [B, P] = (scale / s2) (y - f - m)
        if mean_const != 0.0:
            r.sub_(mean_const)
        r.mul_(scale / noise_variance)
    plain = not q_diag and not het and P <= 16
    # (plain: r q_mu^T - 2 c P At in ONE pass over At; it was a K = P GEMM, then an axpy pass behind the products below)
    Atb = ops.lowrank_axpy(-2.0 * c * P, At, r, q_mu) if plain else ops.gemm_nt(r, q_mu)  # r q_mu^T  [B, M]
    if q_diag:                                                                          # + 2c At (sum_p q_p^2 - P) per column
        if het:
            Atb.addcmul_(At * (2.0 * cvec)[:, None], ((q_sqrt * q_sqrt).sum(1) - P)[None, :])
        else:
            Atb.addcmul_(At, (2.0 * c) * ((q_sqrt * q_sqrt).sum(1) - P)[None, :])
    elif het:
        Wc = W * cvec[None, :, None]                                                    # rows of W_p scaled by c_b
        for p in range(P):
            ops.gemm_nt(Wc[p], Lq[p], alpha=2.0, beta=1.0, C=Atb, b_tri=2)
        Atb.addcmul_(At, cvec[:, None], value=-2.0 * P)
    else:
        for p in range(P):                                                              # + 2c W_p Lq_p^T (Lq_p lower: b_tri 2)
            ops.gemm_nt(W[p], Lq[p], alpha=2.0 * c, beta=1.0, C=Atb, b_tri=2)
        if not plain:
            Atb.add_(At, alpha=-2.0 * c * P)                                            # - 2 c P At
    A = ops.transpose(At)                                                               # [M, B]
    Kfu_bar = ops.gemm_nt(Atb, LinvT, b_tri=1)                                          # At_bar Lm^-1  [B, M]
    Kuf_bar = ops.transpose(Kfu_bar)                                                    # [M, B]

    # branch_q in two parts: its HBM-bound preparation (At^T r, the transposes of W) and its chip-filling GEMMs
    def branch_q_prep():
        g_mu = splitk_gemm_nt(A, r.t().contiguous()) - kl_weight * q_mu                # At^T r - q_mu
        if q_diag:
            return g_mu, None, None
        Wg, ag = (Wc, 2.0) if het else (W, 2.0 * c)
        return g_mu, [ops.transpose(Wg[p]) for p in range(P)], ag

    def branch_q_products(WgT, ag):
        if q_diag:   # d/dq = 2c colsum(At^2) q - (q - 1/q)   (KL of a diagonal q: kullback_leiblers.py:131-133,146-148)
            if het:
                colsq2c = 2.0 * ((At * At) * cvec[:, None]).sum(0)                      # 2 sum_b c_b At[b, m]^2
                return colsq2c[:, None] * q_sqrt - kl_weight * (q_sqrt - 1.0 / q_sqrt)
            colsq = ops.row_stats(A)[0]
            return (2.0 * c) * colsq[:, None] * q_sqrt - kl_weight * (q_sqrt - 1.0 / q_sqrt)
        g = torch.stack([splitk_gemm_nt(A, WgT[p], c_lower=True, alpha=ag) for p in range(P)]) if P > 1 else \
            splitk_gemm_nt(A, WgT[0], c_lower=True, alpha=ag).unsqueeze(0)
        g.sub_(Lq, alpha=kl_weight)                                                     # 2c tril(At^T W_p) - Lq_p
        g.diagonal(dim1=1, dim2=2).add_(kl_weight / Lq.diagonal(dim1=1, dim2=2))
        return g

    side = _side_stream(dev) if (OVERLAP_BRANCHES and Z.is_cuda) else None
    LT = ops.transpose(L, mode=1)
    if side is not None:
        # The side branch reads A, W, r, Lq, q_mu (allocated on the main stream) and allocates its split-K partials from
        # the side stream's pool (~270 MB per latent at M = 2048).  The join below sits in a `finally`: if anything on the
        # main branch raises, the function still waits for the side stream before its locals go back to the allocator.
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            g_qmu, WgT, ag = branch_q_prep()
        g_qs_box = []

        # The main branch is the LONGER one (Lm_bar, three dependent M^3 products, two kernel adjoints: ~2.5 ms against ~1.6) and
        # its short HBM-bound kernels starve while a chip-filling GEMM of the other stream has workgroups waiting for a slot: the
        # combine / transpose pair behind the Lm_bar product took 298 + 239 us beside the side branch's GEMM, 41 + 11 us alone
        # (profiles/r06_train_timeline_before.txt).  So the side branch's GEMMs wait until that preparation is enqueued; from
        # there on the main branch only runs under-filled products that are bound by their longest tile, not by the CUs they get.
        def release_side():
            with torch.cuda.stream(side):
                if GATE_SIDE_BRANCH:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                g_qs_box.append(branch_q_products(WgT, ag))
        if not GATE_SIDE_BRANCH:
            release_side()
            release_side = None
    else:
        release_side = None
    try:
        Lbar = splitk_gemm_nt(Kuf_bar, A, c_lower=True, alpha=-1.0)            # -tril(Kfu_bar^T At)
        Kuu_bar = cholesky_adjoint(LT, LinvT, Lbar, before_products=release_side)
        dkd = spec.dkdiag()                                                             # Knn = kdiag in every fvar
        csum = cvec.sum() * P if het else c * B * P
        one_kernel = spec.n == 1 and spec.tree is None and spec.cols[0] is None and not het
        if one_kernel:
            # one stationary kernel over all input columns: the second adjoint ADDS its variance / lengthscale / Z gradients to the
            # first one's in its own tail launch, and the Knn term of d/dvariance rides along (no elementwise launches at all)
            fam1, var1, ls1 = spec.members[0]
            packed = stationary_kernel_adjoint(Z, Xb, Kuf_bar, variance=var1, lengthscales=ls1, symmetric=False, family=fam1,
                                               dvar_add=csum * dkd[0], return_packed=True)
            small_g, Zbar = stationary_kernel_adjoint(Z, Z, Kuu_bar, variance=var1, lengthscales=ls1, symmetric=True, family=fam1,
                                                      packed_into=packed, return_packed=True)
        else:
            dv1, dl1, Zb1 = spec.adjoint(Z, Xb, Kuf_bar, symmetric=False)
            dv2, dl2, Zb2 = spec.adjoint(Z, Z, Kuu_bar, symmetric=True)
    finally:
        if side is not None:
            main.wait_stream(side)
    if side is not None:
        g_qs = g_qs_box[0]
        g_qmu.record_stream(main)
        g_qs.record_stream(main)
    else:
        g_qmu, WgT, ag = branch_q_prep()
        g_qs = branch_q_products(WgT, ag)
    if one_kernel:
        g_var = small_g[0:1]
        g_ls = small_g[1:] if not (np.ndim(ls1) == 0 or np.size(ls1) == 1) else small_g[1:].sum().reshape(1)
    else:
        g_var, g_ls = spec.pack([a + b + csum * dk for a, b, dk in zip(dv1, dv2, dkd)], [a + b for a, b in zip(dl1, dl2)])
        Zbar = Zb1 + Zb2
    if het:
        # dF/d sigma_n^2 = scale sum_p (-1 / (2 s2_n) + ((y - f)^2 + fvar) / (2 s2_n^2)): one entry per row (likelihood parameters
        # are reached through Gaussian.noise_param_grads)
        fvar = (spec.kdiag() - s0)[:, None] + ssq.t()
        resid = Yb - fmean - mean_const
        g_noise = scale * (-0.5 * P / nv + 0.5 * (resid * resid + fvar).sum(1) / (nv * nv))
    else:
        # sum_bp ((y - f)^2 + fvar) recovered from the forward value:  ve = B P k0 - Q / (2 s2)
        #   d/ds2 = scale (-B P / (2 s2) + Q / (2 s2^2)) = (scale / s2) (B P (k0 - 1/2) - ve)          (two launches)
        k0 = -0.5 * LOG2PI - 0.5 * float(np.log(noise_variance))
        g_noise = torch.mul(ve, -scale / noise_variance).add_(scale / noise_variance * B * P * (k0 - 0.5)).reshape(1)
    grads = {"variance": g_var, "lengthscales": g_ls, "noise_variance": g_noise,
             "Z": Zbar, "q_mu": g_qmu, "q_sqrt": g_qs, "mean_const": r.sum().reshape(1)}
    return F, grads, info


def gpr_lml_and_grad(X: torch.Tensor, Y: torch.Tensor, *, variance: float = None, lengthscales=None, noise_variance: float,
                     mean_const: float = 0.0, family: str = "SquaredExponential", kernel_spec: "KernelSpec" = None
                     ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """GPR.log_marginal_likelihood (gpr.py:91-107) and its gradient w.r.t. {variance, lengthscales, noise_variance,
    mean_const} for a SquaredExponential kernel -- what `optimizers/scipy.py:322-331` asks TF autodiff for.

        L = chol(K + s2 I),  alpha = L^-1 (Y - m),  LML = -0.5 |alpha|^2 - 0.5 N P log 2pi - P sum log diag L
        K_bar = 0.5 (beta beta^T - P K^-1),  beta = K^-1 (Y - m)        d/ds2 = tr K_bar,   d/dm = sum beta

    The trapezoid is [K + s2 I ; (Y - m)^T ; I]: the identity rows return L^-T, so K^-1 = L^-T L^-1 and beta are GEMMs
    (no triangular solve), and the kernel-parameter contraction is the same G = K_bar .* K pass as for the SVGP."""
    N, D = X.shape
    P = Y.shape[1]
    dev = X.device
    spec = kernel_spec if kernel_spec is not None else KernelSpec.single(variance, lengthscales, family)
    spec.warm(dev, D)   # (device copies of the lengthscales BEFORE the first kernel is enqueued: see ls_device)
    T = torch.empty((N + P + N, N), dtype=torch.float64, device=dev)
    het = torch.is_tensor(noise_variance) and noise_variance.dim() >= 1   # one variance per data row (heteroskedastic likelihood)
    if het:
        spec.build(X, None, T[:N])
        ops.diag_add_(T[:N], noise_variance)                                            # add_likelihood_noise_cov, model_utils.py:46-50
    else:
        spec.build(X, None, T[:N], diag_add=noise_variance)
    T[N:N + P] = (Y - mean_const).t()
    invd, info = ops.potrf_(T, N, zero_upper=True, identity_rows=True)                  # (the last N rows: I -> L^-T, N^3/3)
    L, alphat, LinvT = T[:N], T[N:N + P], T[N + P:]
    a2 = ops.row_stats(alphat)[0].sum()
    lml = -0.5 * a2 - 0.5 * N * P * LOG2PI - P * torch.log(torch.diagonal(L)).sum()
    # beta^T = alpha^T L^-1  [P, N];  K^-1 = L^-T (L^-T)^T, lower tiles only
    betat = ops.gemm_nt(alphat, LinvT, b_tri=1)
    Kbar = ops.gemm_nt(LinvT, LinvT, b_tri=1, c_lower=True, a_tri=1)               # (both factors upper: N^3/3 multiply-adds)
    beta = betat.t().contiguous()                                                       # [N, P]
    ops.gemm_nt(beta, beta, alpha=0.5, beta=-0.5 * P, C=Kbar, c_lower=True)             # 0.5 beta beta^T - 0.5 P K^-1
    low = torch.tril(Kbar)
    Kbar = low + torch.tril(low, -1).t()                                                # symmetric, full
    dvs, dlss, _ = spec.adjoint(X, X, Kbar, symmetric=True)
    dvar, dls = spec.pack(dvs, dlss)
    # d LML / d sigma_n^2 = Kbar_nn: their sum for a constant noise variance, the vector for a heteroskedastic likelihood
    grads = {"variance": dvar, "lengthscales": dls,
             "noise_variance": torch.diagonal(Kbar).clone() if het else torch.diagonal(Kbar).sum().reshape(1),
             "mean_const": betat.sum().reshape(1)}
    return lml.reshape(1), grads, info


def sgpr_elbo_and_grad(Z: torch.Tensor, X: torch.Tensor, Y: torch.Tensor, *, variance: float = None, lengthscales=None,
                       noise_variance: float, jitter: float, mean_const: float = 0.0,
                       family: str = "SquaredExponential", group=None, sharded: bool = False,
                       num_data: Optional[int] = None, kernel_spec: "KernelSpec" = None
                       ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """SGPR.elbo (sgpr.py:181-290) and its gradient w.r.t. {variance, lengthscales, noise_variance, Z, mean_const}.
    With At = Kfu Lm^-T, S = At^T At, a = At^T err, q = |At|^2, e2 = |err|^2, B = I + S / s2, w = B^-1 a / s2:

        F     = -N P log(2 pi)/2 - P (logdet B / 2 + N log(s2)/2 + (N var - q) / (2 s2)) - e2 / (2 s2) + a^T w / (2 s2)
        B_bar = -(P B^-1 + w w^T) / 2,   S_bar = B_bar / s2,   a_bar = w / s2,   q_bar = P / (2 s2)
        At_bar = At (2 S_bar + 2 q_bar I) + err a_bar^T        -> then exactly the SVGP chain: Kfu_bar, Lm_bar, Kuu_bar, kernel

    `sharded=True` (NEW: the reference is single process): (X, Y) is THIS rank's row shard (possibly empty), `num_data`
    the global row count.  The forward statistics (S, a, e2, q) are summed over `group` as in models/sgpr.py; everything
    of size M x M after that is replicated.  The reverse pass is linear in what the shards contribute -- At_bar, Kfu_bar,
    -tril(Kfu_bar^T At) and the Kuf part of the kernel adjoint are per-shard -- so ONE more all-reduce of the packed
    (Lm_bar part [M, M], d/dvariance, d/dlengthscales, Z_bar, d/dmean) gives every rank the complete gradient: two
    collectives of M^2 + O(M) doubles per evaluation, none proportional to the data."""
    M, D = Z.shape
    n, P = Y.shape
    dev = Z.device
    # het: `noise_variance` a tensor [n] of one sigma_n^2 per row of THIS shard (a heteroskedastic Gaussian likelihood, sgpr.py:207-211).
    # With w_n = 1 / sigma_n^2 the statistics become S = At^T W At, a = At^T W err, e2 = sum w err^2, q = sum w |At_n|^2 and the bound is
    # the constant-noise one with s2 = 1 plus -P/2 (sum log sigma_n^2 + var sum w_n); At_bar and d/dmean pick up w_n per row, and
    #     dF/dw_n = (a_n^T (2 S_bar + 2 q_bar I) a_n) / 2 + sum_p (a_n . a_bar_p) err_np - sum_p err_np^2 / 2 - P var / 2,
    #     dF/dsigma_n^2 = -w_n^2 dF/dw_n - P w_n / 2        (one entry per row; grads["noise_variance"] is then [n], local to the shard)
    het = torch.is_tensor(noise_variance) and noise_variance.dim() >= 1
    if het:
        wn = (1.0 / noise_variance.reshape(-1)).contiguous()
        swn = torch.sqrt(wn)
        s2 = 1.0
    else:
        s2 = float(noise_variance)
    N = int(num_data) if num_data is not None else n
    if not sharded and N != n:
        raise ValueError("num_data differs from the number of rows of a model that is not sharded")
    # (kernel_spec: a Sum / Product of stationary kernels, members possibly over different input columns -- round 5)
    spec = kernel_spec if kernel_spec is not None else KernelSpec.single(variance, lengthscales, family)
    spec.warm(dev, D)   # (device copies of the lengthscales BEFORE the first kernel is enqueued: see ls_device)
    kdiag = spec.kdiag()
    nmem = spec.n
    nlsm = [int(np.size(ls)) for _, _, ls in spec.members]     # lengthscale entries per member (1 = isotropic)

    def all_reduce(t):
        if sharded:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    eye = torch.eye(M, dtype=torch.float64, device=dev)
    T = torch.empty((M + n + M, M), dtype=torch.float64, device=dev)
    spec.build(Z, None, T[:M], diag_add=jitter)
    if n:
        spec.build(X, Z, T[M:M + n])
    _, info = ops.potrf_(T, M, zero_upper=True, identity_rows=True)
    L, At, LinvT = T[:M], T[M:M + n], T[M + n:]
    err = (Y - mean_const).contiguous()
    stats = torch.zeros(M * M + M * P + 4, dtype=torch.float64, device=dev)
    if n:
        A = ops.transpose(At)                                                           # [M, n]
        if het:   # rows scaled by 1 / sigma_n (elementwise glue on [n, M] / [n, P])
            Aw = ops.transpose((At * swn[:, None]).contiguous())
            errw = (err * swn[:, None]).contiguous()
            stats[-4], stats[-3] = -torch.log(wn).sum(), wn.sum()                      # sum log sigma_n^2, sum 1 / sigma_n^2
        else:
            Aw, errw = A, err
        stats[:M * M] = torch.tril(splitk_gemm_nt(Aw, Aw, c_lower=True)).reshape(-1)
        stats[M * M:M * M + M * P] = splitk_gemm_nt(Aw, errw.t().contiguous()).reshape(-1)
        stats[-2], stats[-1] = ops.sumsq(errw)[0], (ops.sumsq(Aw)[0] if het else ops.sumsq(At)[0])
    all_reduce(stats)                                                                   # ---- exchange 1: S, a, e2, q (+ the two noise sums)
    Slow = stats[:M * M].reshape(M, M)
    S = Slow + torch.tril(Slow, -1).t()
    a = stats[M * M:M * M + M * P].reshape(M, P)
    e2, q = stats[-2], stats[-1]
    sum_log_s2, sum_w = (stats[-4], stats[-3]) if het else (N * float(np.log(s2)), N / s2)
    T2 = torch.empty((2 * M + P, M), dtype=torch.float64, device=dev)
    T2[:M] = S / s2 + eye
    T2[M:M + P] = a.t() / s2
    _, info2 = ops.potrf_(T2, M, zero_upper=True, identity_rows=True)
    LB, ct, LBinvT = T2[:M], T2[M:M + P], T2[M + P:]
    half_logdet_b = ops.sum_log_diag(LB)[0]
    F = (-0.5 * N * P * LOG2PI - P * (half_logdet_b + 0.5 * sum_log_s2 + 0.5 * (kdiag * sum_w - q / s2))
         - 0.5 * (e2 / s2 - ops.sumsq(ct)[0]))
    # ---- backward: the M x M tail (replicated)
    Binv = ops.gemm_nt(LBinvT, LBinvT, b_tri=1, a_tri=1)                                # B^-1 = LB^-T LB^-1
    wt = ops.gemm_nt(ct.contiguous(), LBinvT, b_tri=1)                                  # w^T = c^T LB^-1  [P, M]
    w = wt.t().contiguous()                                                             # [M, P]
    Bbar = -0.5 * P * Binv
    ops.gemm_nt(w, w, alpha=-0.5, beta=1.0, C=Bbar)                                     # - w w^T / 2
    Ssym = (2.0 / s2) * Bbar + (P / s2) * eye                                           # 2 S_bar + 2 q_bar I
    abar = w / s2
    # ---- this shard's rows
    # packed per-shard sums: [Lm_bar part (M x M), d/dvariance per member, d/dlengthscales per member, Z_bar (M x D), d/dmean]
    nl_tot = sum(nlsm)
    part = torch.zeros(M * M + nmem + nl_tot + M * D + 1, dtype=torch.float64, device=dev)
    o = M * M
    g_noise_rows = None
    if n:
        Atb = ops.gemm_nt(err, abar)                                                    # err a_bar^T  [n, M]
        if het:
            G = ops.gemm_nt(At, Ssym)                                                   # At (2 S_bar + 2 q_bar I)
            Au = ops.gemm_nt(At, abar.t().contiguous())                                 # (a_n . a_bar_p)  [n, P]
            dF_dw = 0.5 * (G * At).sum(1) + (Au * err).sum(1) - 0.5 * (err * err).sum(1) - 0.5 * P * kdiag
            g_noise_rows = -(wn * wn) * dF_dw - 0.5 * P * wn
            Atb = (Atb + G) * wn[:, None]
        else:
            ops.gemm_nt(At, Ssym, alpha=1.0, beta=1.0, C=Atb)                           # + At (2 S_bar + 2 q_bar I)
        Kfu_bar = ops.gemm_nt(Atb, LinvT, b_tri=1)
        Kuf_bar = ops.transpose(Kfu_bar)
        part[:M * M] = torch.tril(splitk_gemm_nt(Kuf_bar, A, c_lower=True, alpha=-1.0)).reshape(-1)
        dv1, dl1, Zb1 = spec.adjoint(Z, X, Kuf_bar, symmetric=False)
        part[o:o + nmem] = torch.cat(dv1)
        part[o + nmem:o + nmem + nl_tot] = torch.cat([d.reshape(-1) for d in dl1])
        part[o + nmem + nl_tot:o + nmem + nl_tot + M * D] = Zb1.reshape(-1)
        part[-1] = ((err - Au) * wn[:, None]).sum() if het else (err / s2 - ops.gemm_nt(At, abar.t().contiguous())).sum()
    all_reduce(part)                                                                    # ---- exchange 2: the shards' sums
    Lbar = part[:o].reshape(M, M)
    dv1 = [part[o + i].reshape(1) for i in range(nmem)]
    offs = np.concatenate([[0], np.cumsum(nlsm)])
    dl1 = [part[o + nmem + offs[i]:o + nmem + offs[i + 1]] for i in range(nmem)]
    Zb1, g_mean = part[o + nmem + nl_tot:o + nmem + nl_tot + M * D].reshape(M, D), part[-1]
    Kuu_bar = cholesky_adjoint(ops.transpose(L, mode=1), LinvT, Lbar)
    dv2, dl2, Zb2 = spec.adjoint(Z, Z, Kuu_bar, symmetric=True)
    g_var, g_ls = spec.pack([a + b - 0.5 * P * sum_w * dk for a, b, dk in zip(dv1, dv2, spec.dkdiag())],
                            [a + b for a, b in zip(dl1, dl2)])
    if het:
        g_noise = g_noise_rows if g_noise_rows is not None else torch.zeros(0, dtype=torch.float64, device=dev)
    else:
        wa, ww = (w * a).sum(), (w * w).sum()
        g_noise = (-P * (-0.5 * (M - torch.diagonal(Binv).sum()) / s2 + 0.5 * N / s2 - 0.5 * (N * kdiag - q) / s2 ** 2)
                   + 0.5 * e2 / s2 ** 2 - 0.5 * wa / s2 ** 2 - 0.5 * ww / s2).reshape(1)
    status = torch.maximum(info, info2)
    grads = {"variance": g_var, "lengthscales": g_ls, "noise_variance": g_noise, "Z": Zb1 + Zb2,
             "mean_const": g_mean.reshape(1)}
    return F.reshape(1), grads, status


def svgp_elbo_and_grad_unwhitened(Z: torch.Tensor, Xb: torch.Tensor, Yb: torch.Tensor, q_mu: torch.Tensor,
                                  q_sqrt: torch.Tensor, *, variance: float = None, lengthscales=None, noise_variance: float,
                                  jitter: float, scale: float = 1.0, mean_const: float = 0.0, kl_weight: float = 1.0,
                                  family: str = "SquaredExponential", kernel_spec: "KernelSpec" = None
                       ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """The `whiten=False` SVGP: q(u) = N(q_mu, Lq Lq^T) on u itself (conditionals/util.py:137-139, kullback_leiblers.py:
    98-165 with K = Kuu).  With Linv = Lm^-1 (from the identity rows of the trapezoid):

        forward   A2t = At Linv,  fmean = A2t q_mu,  W_p = A2t Lq_p,  fvar = var - rowsum(At^2) + rowsum(W_p^2)
                  KL = 0.5 sum_p (|Linv q_mu_p|^2 - M - log|Lq_p|^2 + |Linv Lq_p|_F^2) + P sum log diag Lm
        backward  A2t_bar = r q_mu^T + 2c sum_p W_p Lq_p^T,   At_bar = -2cP At + A2t_bar Linv^T,
                  Linv_bar = tril(At^T A2t_bar) - k (alpha q_mu^T + sum_p V_p Lq_p^T)      (alpha = Linv q_mu, V_p = Linv Lq_p)
                  Lm_bar   = -tril(Kfu_bar^T At) - tril(Linv^T Linv_bar Linv^T) - k P diag(1 / Lm)
    and then the same Cholesky / kernel adjoints as the whitened path.  q_diag (q_sqrt [M, P] of standard deviations;
    conditionals/util.py:147-149, kullback_leiblers.py:131-133, 146-152) is the same algebra with Lq_p = diag(q_p): W_p =
    A2t diag(q_p) is never formed (row statistics with weights), sum_p V_p Lq_p^T = Linv diag(sum_p q_p^2), and the trace
    term is sum_m (Kuu^-1)_mm sum_p q_mp^2 with diag(Kuu^-1) = column sums of Linv^2.  Reached through `SVGP.elbo_and_grad` /
    `SVGPTrainer` when `whiten=False`; parity vs the autograd oracle on the emulated primitives
    (tests/test_gradients_cpu.py) and on the GPU (tests/test_gpu_gradients.py)."""
    M, D = Z.shape
    B = Xb.shape[0]
    P = q_mu.shape[1]
    q_diag = q_sqrt.dim() == 2
    if tuple(q_sqrt.shape) not in ((P, M, M), (M, P)):
        raise ValueError("svgp_elbo_and_grad_unwhitened needs q_sqrt [P, M, M] or, for q_diag, [M, P]")
    dev = Z.device
    # (kernel_spec: a Sum / Product of stationary kernels, members possibly over different input columns -- round 5)
    spec = kernel_spec if kernel_spec is not None else KernelSpec.single(variance, lengthscales, family)
    spec.warm(dev, D)   # (device copies of the lengthscales BEFORE the first kernel is enqueued: see ls_device)
    k = float(kl_weight)
    T = torch.empty((M + B + M, M), dtype=torch.float64, device=dev)
    spec.build(Z, None, T[:M], diag_add=jitter)
    spec.build(Xb, Z, T[M:M + B])
    _, info = ops.potrf_(T, M, zero_upper=True, identity_rows=True)
    L, At, LinvT = T[:M], T[M:M + B], T[M + B:]
    Linv = ops.transpose(LinvT)                                                         # lower
    A2t = ops.gemm_nt(At, LinvT, b_tri=1)                                               # At Linv   (util.py:139)
    s0 = ops.row_stats(At)[0]
    alphat = ops.gemm_nt(q_mu.t().contiguous(), Linv, b_tri=2)                          # (Linv q_mu)^T  [P, M]
    if q_diag:
        qd = q_sqrt.contiguous()
        s = (qd * qd).sum(1)                                                            # sum_p q_mp^2  [M]
        _, fmean, ssq = ops.row_stats(A2t, V=q_mu, W=qd)                                # A2t q_mu, sum_m A2t^2 q_mp^2
        kinv_diag = ops.row_stats(LinvT)[0]                                             # diag(Kuu^-1) = rowsum((Lm^-T)^2)
        kl = 0.5 * (ops.sumsq(alphat)[0] - M * P - torch.log(qd * qd).sum() + (kinv_diag * s).sum()) \
            + P * ops.sum_log_diag(L)[0]
    else:
        Lq = _tril(q_sqrt)
        LqT = ops.transpose(q_sqrt, mode=1)
        _, fmean, _ = ops.row_stats(A2t, V=q_mu, want_sumsq=False)
        W = ops.gemm_nt(A2t, LqT, b_tri=1)                                              # [P, B, M]
        ssq = torch.stack([ops.row_stats(W[p])[0] for p in range(P)])
        V = ops.gemm_nt(Linv, LqT, b_tri=1, a_tri=2)                                    # [P, M, M]: V_p = Linv Lq_p
        if V.dim() == 2:
            V = V.unsqueeze(0)
        kl = 0.5 * (ops.sumsq(alphat)[0] - M * P - torch.log(Lq.diagonal(dim1=1, dim2=2) ** 2).sum()
                    + sum(ops.sumsq(V[p])[0] for p in range(P))) + P * ops.sum_log_diag(L)[0]
    ve, _ = ops.gaussian_varexp_sum(Yb, fmean, s0=s0, ssq=ssq, knn=[spec.kdiag()], noise_variance=noise_variance,
                                    mean_const=mean_const)
    F = scale * ve - k * kl
    # ---- backward
    # het: one noise variance per row (a heteroskedastic Gaussian likelihood): dF/dfvar is a per-row vector c_b, applied as a row
    # scaling of the factors the scalar multiplied (same treatment as the whitened pass); the scalar path is unchanged
    het = torch.is_tensor(noise_variance) and noise_variance.dim() >= 1
    if het:
        nv = noise_variance.reshape(-1)
        cvec = (-0.5 * scale) / nv
        c = None
        r = (scale / nv)[:, None] * (Yb - fmean - mean_const)
    else:
        c = -0.5 * scale / noise_variance
        r = (scale / noise_variance) * (Yb - fmean - mean_const)
    A2tb = ops.gemm_nt(r, q_mu)
    Wc = None
    if q_diag:
        if het:
            A2tb.addcmul_(A2t * (2.0 * cvec)[:, None], s[None, :])
        else:
            A2tb.addcmul_(A2t, (2.0 * c) * s[None, :])
    elif het:
        Wc = W * cvec[None, :, None]                                                    # rows of W_p scaled by c_b
        for p in range(P):
            ops.gemm_nt(Wc[p], Lq[p], alpha=2.0, beta=1.0, C=A2tb, b_tri=2)
    else:
        for p in range(P):
            ops.gemm_nt(W[p], Lq[p], alpha=2.0 * c, beta=1.0, C=A2tb, b_tri=2)
    Atb = ops.gemm_nt(A2tb, Linv, b_tri=2)                                              # A2t_bar Linv^T
    if het:
        Atb.addcmul_(At, cvec[:, None], value=-2.0 * P)
    else:
        Atb.add_(At, alpha=-2.0 * c * P)
    A = ops.transpose(At)
    A2 = ops.transpose(A2t)
    Kfu_bar = ops.gemm_nt(Atb, LinvT, b_tri=1)
    Kuf_bar = ops.transpose(Kfu_bar)
    # q(u) gradients: data part through A2t, KL part through Kuu^-1
    Kinv_qmu_t = ops.gemm_nt(alphat, LinvT, b_tri=1)                                    # (Linv^T alpha)^T  [P, M]
    g_qmu = splitk_gemm_nt(A2, r.t().contiguous()) - k * Kinv_qmu_t.t()
    if q_diag:   # d/dq_mp = 2c colsum(A2t^2)_m q_mp - k ((Kuu^-1)_mm q_mp - 1 / q_mp)
        colsq2c = 2.0 * ((A2t * A2t) * cvec[:, None]).sum(0) if het else (2.0 * c) * ops.row_stats(A2)[0]
        g_qs = colsq2c[:, None] * qd - k * (kinv_diag[:, None] * qd - 1.0 / qd)
    else:
        Wg, ag = (Wc, 2.0) if het else (W, 2.0 * c)
        g_qs = torch.stack([splitk_gemm_nt(A2, ops.transpose(Wg[p]), c_lower=True, alpha=ag) for p in range(P)])
        for p in range(P):
            KinvLq = ops.gemm_nt(LinvT, ops.transpose(V[p], mode=1), b_tri=1, a_tri=1)  # Linv^T V_p = Kuu^-1 Lq_p (V_p lower)
            g_qs[p] -= k * torch.tril(KinvLq)
        g_qs.diagonal(dim1=1, dim2=2).add_(k / Lq.diagonal(dim1=1, dim2=2))
    # Linv_bar (lower) and its pull-back to Lm
    Linv_bar = splitk_gemm_nt(A, ops.transpose(A2tb), c_lower=True)         # tril(At^T A2t_bar)
    Linv_bar -= k * torch.tril(ops.gemm_nt(alphat.t().contiguous(), q_mu))              # alpha q_mu^T
    if q_diag:
        Linv_bar -= k * torch.tril(Linv) * s[None, :]                                   # Linv diag(sum_p q_p^2)
    for p in range(0 if q_diag else P):
        Linv_bar -= k * torch.tril(ops.gemm_nt(V[p], Lq[p], b_tri=2))                   # V_p Lq_p^T
    X1 = ops.gemm_nt(LinvT, ops.transpose(Linv_bar, mode=1), b_tri=1, a_tri=1)                   # Linv^T Linv_bar (Linv_bar lower)
    X2 = ops.gemm_nt(X1, Linv, b_tri=2)                                                 # (.) Linv^T
    Lbar = splitk_gemm_nt(Kuf_bar, A, c_lower=True, alpha=-1.0) - torch.tril(X2)
    Lbar.diagonal().sub_(k * P / L.diagonal())
    Kuu_bar = cholesky_adjoint(ops.transpose(L, mode=1), LinvT, Lbar)
    dv1, dl1, Zb1 = spec.adjoint(Z, Xb, Kuf_bar, symmetric=False)
    dv2, dl2, Zb2 = spec.adjoint(Z, Z, Kuu_bar, symmetric=True)
    csum = cvec.sum() * P if het else c * B * P
    g_var, g_ls = spec.pack([a + b + csum * dk for a, b, dk in zip(dv1, dv2, spec.dkdiag())],
                            [a + b for a, b in zip(dl1, dl2)])
    if het:   # dF/d sigma_n^2 per row (likelihood parameters are reached through Gaussian.noise_param_grads)
        fvar = (spec.kdiag() - s0)[:, None] + ssq.t()
        resid = Yb - fmean - mean_const
        g_noise = scale * (-0.5 * P / nv + 0.5 * (resid * resid + fvar).sum(1) / (nv * nv))
    else:
        k0 = -0.5 * LOG2PI - 0.5 * float(np.log(noise_variance))
        Q = 2.0 * noise_variance * (B * P * k0 - ve)
        g_noise = (scale * (-0.5 * B * P / noise_variance + 0.5 * Q / noise_variance ** 2)).reshape(1)
    grads = {"variance": g_var, "lengthscales": g_ls, "noise_variance": g_noise,
             "Z": Zb1 + Zb2, "q_mu": g_qmu, "q_sqrt": g_qs, "mean_const": r.sum().reshape(1)}
    return F, grads, info
