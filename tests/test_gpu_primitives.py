"""GPU parity tests of the libgpk primitives (through the C-ABI) against the NumPy/SciPy oracle.

Tolerances are stated per test; fp64 throughout.  Inputs are asymmetric on purpose (transpose-detecting).
"""
import os

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as orc  # noqa: E402  (test-side checker only)


def _t(x):
    from gpflow_amd import ops
    return ops.to_device(x)


def _spd(rng, n, d=3, noise=0.1):
    X = rng.normal(size=(n, d))
    return X, orc.rbf_K(X, variance=1.3, lengthscales=0.9 * np.sqrt(d)) + noise * np.eye(n)


@pytest.mark.parametrize("family", ["SquaredExponential", "Matern12", "Matern32", "Matern52"])
@pytest.mark.parametrize("n1,n2,d,ard", [(70, 133, 3, True), (64, 64, 8, False), (1, 5, 1, False), (257, 300, 16, True)])
def test_kernel_matrix_cross(gpu, family, n1, n2, d, ard):
    from gpflow_amd import ops
    rng = np.random.default_rng(0)
    X1, X2 = rng.normal(size=(n1, d)), rng.normal(size=(n2, d))
    ls = (0.7 + 0.1 * np.arange(d)) if ard else 0.8
    K = ops.kernel_matrix(_t(X1), _t(X2), variance=2.3, lengthscales=ls, family=family).cpu().numpy()
    ref = orc.stationary_K(family, X1, X2, variance=2.3, lengthscales=ls)
    np.testing.assert_allclose(K, ref, rtol=0, atol=2e-14)


@pytest.mark.parametrize("n,d", [(5, 2), (64, 8), (200, 3), (513, 8)])
def test_kernel_matrix_sym(gpu, n, d):
    from gpflow_amd import ops
    rng = np.random.default_rng(1)
    X = rng.normal(size=(n, d))
    K = ops.kernel_matrix(_t(X), None, variance=1.7, lengthscales=1.4, diag_add=0.25).cpu().numpy()
    ref = orc.rbf_K(X, variance=1.7, lengthscales=1.4) + 0.25 * np.eye(n)
    np.testing.assert_allclose(K, ref, rtol=0, atol=2e-14)
    Kl = ops.kernel_matrix(_t(X), None, variance=1.7, lengthscales=1.4, diag_add=0.25, lower_only=True).cpu().numpy()
    np.testing.assert_allclose(np.tril(Kl), np.tril(ref), rtol=0, atol=2e-14)


@pytest.mark.parametrize("m,n,k", [(128, 128, 16), (130, 70, 37), (257, 129, 128), (64, 300, 513), (5, 3, 2), (1000, 64, 64)])
def test_gemm_nt(gpu, m, n, k):
    from gpflow_amd import ops
    rng = np.random.default_rng(2)
    A, B, C = rng.normal(size=(m, k)), rng.normal(size=(n, k)), rng.normal(size=(m, n))
    out = ops.gemm_nt(_t(A), _t(B), alpha=-0.7, beta=1.3, C=_t(C)).cpu().numpy()
    ref = -0.7 * A @ B.T + 1.3 * C
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12 * max(1, k))


def test_gemm_nt_identity_asymmetric(gpu):
    """A = I against an asymmetric B catches row/col swaps of the MFMA D layout."""
    from gpflow_amd import ops
    n = 128
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)
    out = ops.gemm_nt(_t(np.eye(n)), _t(B)).cpu().numpy()
    np.testing.assert_array_equal(out, B.T)


def test_gemm_nt_tri_and_batch(gpu):
    from gpflow_amd import ops
    rng = np.random.default_rng(3)
    m, n, k = 300, 256, 256
    A = rng.normal(size=(m, k))
    Bu = np.triu(rng.normal(size=(2, n, k)))  # zero for kk < j
    out = ops.gemm_nt(_t(A), _t(Bu), b_tri=1).cpu().numpy()
    np.testing.assert_allclose(out, np.einsum("mk,bnk->bmn", A, Bu), rtol=0, atol=1e-11)
    Bl = np.tril(rng.normal(size=(n, k)))
    out = ops.gemm_nt(_t(A), _t(Bl), b_tri=2).cpu().numpy()
    np.testing.assert_allclose(out, A @ Bl.T, rtol=0, atol=1e-11)
    # lower-only syrk: tiles touching the lower triangle must be exact
    S = rng.normal(size=(384, 64))
    C0 = rng.normal(size=(384, 384))
    out = ops.gemm_nt(_t(S), _t(S), alpha=-1.0, beta=1.0, C=_t(C0), c_lower=True).cpu().numpy()
    ref = C0 - S @ S.T
    np.testing.assert_allclose(np.tril(out), np.tril(ref), rtol=0, atol=1e-11)


@pytest.mark.parametrize("n,extra", [(1, 0), (17, 3), (128, 0), (129, 5), (300, 40), (640, 130), (1100, 257), (5000, 700)])
def test_potrf_trapezoid(gpu, n, extra):
    from gpflow_amd import ops
    rng = np.random.default_rng(4)
    _, K = _spd(rng, n)
    Bm = rng.normal(size=(extra, n))
    T = np.vstack([K, Bm])
    Td = _t(T)
    invd, info = ops.potrf_(Td, n, zero_upper=True)
    ops.check_info(info)
    out = Td.cpu().numpy()
    L = out[:n]
    Lref = np.linalg.cholesky(K)
    np.testing.assert_allclose(L, Lref, rtol=0, atol=5e-13)
    assert np.all(np.triu(L, 1) == 0)
    np.testing.assert_allclose(L @ L.T, K, rtol=0, atol=5e-13)
    if extra:
        ref = sla.solve_triangular(Lref, Bm.T, lower=True).T
        np.testing.assert_allclose(out[n:], ref, rtol=0, atol=1e-11)
    # diagonal-block inverses
    nblk = (n + 127) // 128
    inv = invd.cpu().numpy().reshape(nblk, 128, 128)
    for b in range(nblk):
        j0, j1 = b * 128, min(n, b * 128 + 128)
        D = Lref[j0:j1, j0:j1]
        np.testing.assert_allclose(inv[b][: j1 - j0, : j1 - j0] @ D, np.eye(j1 - j0), rtol=0, atol=1e-11)


@pytest.mark.parametrize("n,extra", [(40, 0), (128, 3), (200, 30), (300, 0), (640, 130), (1152, 700), (2048, 1), (4224, 300)])
def test_potrf_identity_rows(gpu, n, extra):
    """gpk_potrf_inv: the library writes the identity rows and returns L^-T, skipping the rows that are still zero.
    Same values as the dense route (identity rows handed in explicitly; 1e-11: the row count picks the GEMM variant and
    with it the summation order) and exact zeros below the diagonal; checked against the oracle inverse as well."""
    import torch
    from gpflow_amd import ops
    rng = np.random.default_rng(41)
    _, K = _spd(rng, n, noise=0.3)
    Bm = rng.normal(size=(extra, n))
    Td = torch.full((2 * n + extra, n), float("nan"), dtype=torch.float64, device=_t(K).device)   # identity block: garbage in
    Td[:n] = _t(K)
    if extra:
        Td[n:n + extra] = _t(Bm)
    _, info = ops.potrf_(Td, n, zero_upper=True, identity_rows=True)
    ops.check_info(info)
    Tref = _t(np.vstack([K, Bm, np.eye(n)]))
    _, info = ops.potrf_(Tref, n, zero_upper=True)
    ops.check_info(info)
    out, ref = Td.cpu().numpy(), Tref.cpu().numpy()
    assert np.array_equal(out[:n], ref[:n])
    np.testing.assert_allclose(out[n:n + extra], ref[n:n + extra], rtol=0, atol=1e-11)
    LinvT = out[n + extra:]
    assert np.all(np.tril(LinvT, -1) == 0)
    np.testing.assert_allclose(LinvT, ref[n + extra:], rtol=0, atol=1e-11)
    Lref = np.linalg.cholesky(K)
    np.testing.assert_allclose(LinvT.T @ Lref, np.eye(n), rtol=0, atol=1e-10)


def test_potrf_batched(gpu):
    from gpflow_amd import ops
    rng = np.random.default_rng(5)
    n, extra, b = 260, 33, 3
    Ts, Ks, Bs = [], [], []
    for i in range(b):
        _, K = _spd(rng, n, noise=0.1 + 0.1 * i)
        Bm = rng.normal(size=(extra, n))
        Ts.append(np.vstack([K, Bm])); Ks.append(K); Bs.append(Bm)
    Td = _t(np.stack(Ts))
    invd, info = ops.potrf_(Td, n)
    ops.check_info(info)
    out = Td.cpu().numpy()
    for i in range(b):
        Lref = np.linalg.cholesky(Ks[i])
        np.testing.assert_allclose(np.tril(out[i, :n]), Lref, rtol=0, atol=5e-13)
        np.testing.assert_allclose(out[i, n:], sla.solve_triangular(Lref, Bs[i].T, lower=True).T, rtol=0, atol=1e-11)


def test_potrf_not_pd_reports_info(gpu):
    from gpflow_amd import ops, _lib
    K = np.eye(200)
    K[150, 150] = -1.0
    Td = _t(K)
    _, info = ops.potrf_(Td, 200)
    assert int(info.cpu()[0]) == 151
    with pytest.raises(_lib.GpkError):
        ops.check_info(info)


@pytest.mark.parametrize("n,m", [(100, 7), (256, 300), (700, 129)])
def test_trsm_both(gpu, n, m):
    from gpflow_amd import ops
    rng = np.random.default_rng(6)
    _, K = _spd(rng, n)
    L = np.linalg.cholesky(K)
    Bm = rng.normal(size=(m, n))
    Ld = _t(L)
    invd = ops.trtri_blocks(Ld)
    X0 = ops.trsm_(_t(Bm), Ld, invd, trans=0).cpu().numpy()
    np.testing.assert_allclose(X0, sla.solve_triangular(L, Bm.T, lower=True).T, rtol=0, atol=1e-11)
    LT, invdT = ops.transpose_factor(Ld, invd)
    np.testing.assert_array_equal(LT.cpu().numpy(), L.T)
    X1 = ops.trsm_(_t(Bm), LT, invdT, trans=1).cpu().numpy()
    np.testing.assert_allclose(X1, sla.solve_triangular(L.T, Bm.T, lower=False).T, rtol=0, atol=1e-10)


@pytest.mark.parametrize("n,m,batch", [(512, 1024, 1), (384, 1039, 1), (512, 4128, 1), (384, 4119, 1), (256, 4200, 1), (640, 4111, 1),
                                       (512, 1400, 3), (512, 1025, 3)])
def test_trsm_fused_group_solve(gpu, n, m, batch):
    """The fused in-group solve: up to 256 sixteen-row slivers (rows x batch) run on the staged kernel (group_solve_kernel), more on
    the pipelined one (group_solve2_kernel: 32 rows per workgroup, operand quarters through a ring of LDS buffers).  4 / 3 / 2 leaf
    blocks per group, a 4 + 1 split, row counts that are not multiples of 32 or 16, batches of problems on either side of the switch.
    Oracle: scipy triangular solves per problem."""
    import torch
    from gpflow_amd import ops
    rng = np.random.default_rng(61)
    Ls, Bs = [], []
    for _ in range(batch):
        _, K = _spd(rng, n)
        Ls.append(np.linalg.cholesky(K))
        Bs.append(rng.normal(size=(m, n)))
    if batch == 1:
        Ld = _t(Ls[0])
        invd = ops.trtri_blocks(Ld)
        X0 = ops.trsm_(_t(Bs[0]), Ld, invd, trans=0).cpu().numpy()
        np.testing.assert_allclose(X0, sla.solve_triangular(Ls[0], Bs[0].T, lower=True).T, rtol=0, atol=1e-11)
        return
    # batched: the trapezoid entry point with extra rows (batch of factorisations, extra rows solved group by group)
    T = np.stack([np.vstack([L @ L.T, B]) for L, B in zip(Ls, Bs)])
    Td = _t(T)
    invd, info = ops.potrf_(Td, n, zero_upper=True)
    ops.check_info(info)
    out = Td.cpu().numpy()
    for b in range(batch):
        np.testing.assert_allclose(out[b, :n], Ls[b], rtol=0, atol=5e-12)
        np.testing.assert_allclose(out[b, n:], sla.solve_triangular(Ls[b], Bs[b].T, lower=True).T, rtol=0, atol=1e-10)


def test_potrf_many_extra_rows_capped_updates(gpu):
    """n = 1536 with 8192 extra rows: the extra-row stream's K = 512 updates have 512 / 256 tiles, more than the 224 persistent
    workgroups they are capped to, so the remainder of each launch runs as 64 x 64 quarters behind it (launch_fast, "remainder
    round"); the in-group solves are the pipelined kernel at 4 blocks per group.  Oracle: LAPACK Cholesky + triangular solve."""
    from gpflow_amd import ops
    rng = np.random.default_rng(62)
    n, extra = 1536, 8192
    _, K = _spd(rng, n)
    Bm = rng.normal(size=(extra, n))
    Td = _t(np.vstack([K, Bm]))
    invd, info = ops.potrf_(Td, n, zero_upper=True)
    ops.check_info(info)
    out = Td.cpu().numpy()
    Lref = np.linalg.cholesky(K)
    np.testing.assert_allclose(out[:n], Lref, rtol=0, atol=5e-12)
    ref = sla.solve_triangular(Lref, Bm.T, lower=True).T
    np.testing.assert_allclose(out[n:], ref, rtol=0, atol=1e-10)
    # twice the same answer, bit for bit (persistent workgroups + remainder launch: no order depends on timing)
    Td2 = _t(np.vstack([K, Bm]))
    ops.potrf_(Td2, n, zero_upper=True)
    np.testing.assert_array_equal(Td2.cpu().numpy(), out)


def test_row_stats_project_reductions(gpu):
    from gpflow_amd import ops
    rng = np.random.default_rng(7)
    rows, m, P = 333, 200, 3
    At = rng.normal(size=(rows, m))
    V, W = rng.normal(size=(m, P)), rng.uniform(0.5, 1.5, size=(m, P))
    s, mv, wsq = ops.row_stats(_t(At), V=_t(V), W=_t(W))
    np.testing.assert_allclose(s.cpu().numpy(), (At ** 2).sum(1), rtol=1e-13)
    np.testing.assert_allclose(mv.cpu().numpy(), At @ V, rtol=0, atol=1e-12)
    np.testing.assert_allclose(wsq.cpu().numpy(), np.einsum("bk,kp->pb", At ** 2, W ** 2), rtol=1e-13)
    q_sqrt = rng.normal(size=(P, m, m))  # upper part must be ignored (band_part)
    LqT = ops.transpose(_t(q_sqrt), mode=1)
    np.testing.assert_array_equal(LqT.cpu().numpy(), np.transpose(np.tril(q_sqrt), (0, 2, 1)))
    ssq = ops.project(_t(At), LqT).cpu().numpy()
    ref = np.stack([((At @ np.tril(q_sqrt[p])) ** 2).sum(1) for p in range(P)])
    np.testing.assert_allclose(ssq, ref, rtol=1e-12)
    # one At PER latent (gpk_project_batched; SeparateIndependent): rows of a batched trapezoid, i.e. a strided view --
    # bit-identical to P single-latent launches
    T3 = _t(rng.normal(size=(P, 50 + rows, m)))
    ssq_b = ops.project(T3[:, 50:], LqT).cpu().numpy()
    for p in range(P):
        one = ops.project(T3[p, 50:], LqT[p:p + 1]).cpu().numpy()[0]
        np.testing.assert_array_equal(ssq_b[p], one)
        np.testing.assert_allclose(ssq_b[p], ((T3[p, 50:].cpu().numpy() @ np.tril(q_sqrt[p])) ** 2).sum(1), rtol=1e-12)
    with pytest.raises(ValueError):
        ops.project(T3[:2, 50:], LqT)
    # scalar tails
    Y = rng.normal(size=(rows, P)); fmean = rng.normal(size=(rows, P))
    s0 = rng.uniform(0, 0.5, size=rows)
    out, fvar = ops.gaussian_varexp_sum(_t(Y), _t(fmean), s0=_t(s0), ssq=_t(ref), knn=[1.3], noise_variance=0.2,
                                        mean_const=0.1, want_fvar=True)
    fv = 1.3 - s0[:, None] + ref.T
    np.testing.assert_allclose(fvar.cpu().numpy(), fv, rtol=1e-13)
    refsum = orc.gaussian_variational_expectations(fmean + 0.1, fv, Y, 0.2).sum()
    np.testing.assert_allclose(out.cpu().numpy()[0], refsum, rtol=1e-13)
    q_mu = rng.normal(size=(m, P))
    kl = ops.gauss_kl_white(_t(q_mu), _t(q_sqrt)).cpu().numpy()[0]
    np.testing.assert_allclose(kl, orc.gauss_kl(q_mu, q_sqrt), rtol=1e-13)
    kld = ops.gauss_kl_white(_t(q_mu), _t(W)).cpu().numpy()[0]
    np.testing.assert_allclose(kld, orc.gauss_kl(q_mu, W), rtol=1e-13)
    L = np.tril(rng.uniform(0.5, 2.0, size=(m, m)))
    np.testing.assert_allclose(ops.sum_log_diag(_t(L)).cpu().numpy()[0], np.log(np.diag(L)).sum(), rtol=1e-13)
    np.testing.assert_allclose(ops.sumsq(_t(At)).cpu().numpy()[0], (At ** 2).sum(), rtol=1e-13)


@pytest.mark.parametrize("n,d,P", [(50, 1, 1), (512, 2, 1), (700, 8, 3), (4224, 3, 2), (5000, 4, 1)])
def test_fused_gpr_lml(gpu, n, d, P):
    from gpflow_amd import ops
    rng = np.random.default_rng(8)
    X = rng.normal(size=(n, d))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(n, P))
    kw = dict(variance=1.2, lengthscales=0.8 + 0.1 * np.arange(d) if d > 1 else 0.9, noise_variance=0.1)
    out, info = ops.gpr_lml(_t(X), _t(Y), mean_const=0.05, **kw)
    ops.check_info(info)
    ref = orc.gpr_log_marginal_likelihood(X, Y, mean=0.05, **kw)
    np.testing.assert_allclose(out.cpu().numpy()[0], ref, rtol=1e-10)
    if n >= 4096:
        # large n: only the first outer panel's columns are built before the factorisation starts, the rest beside its chain
        # on the bulk stream (potrf.hip, gpk_gpr_lml) -- repeated calls and per-row noise through the same split
        out2, _ = ops.gpr_lml(_t(X), _t(Y), mean_const=0.05, **kw)
        assert float(out2.cpu()[0]) == float(out.cpu()[0])
        nv = rng.uniform(0.05, 0.3, size=n)
        kw["noise_variance"] = nv
        outh, info = ops.gpr_lml(_t(X), _t(Y), mean_const=0.05, **{**kw, "noise_variance": _t(nv)})
        ops.check_info(info)
        np.testing.assert_allclose(outh.cpu().numpy()[0], orc.gpr_log_marginal_likelihood(X, Y, mean=0.05, **kw), rtol=1e-10)


@pytest.mark.parametrize("m,rows,d,P,q_diag", [(20, 50, 1, 2, False), (200, 300, 8, 1, False), (300, 1000, 8, 4, False), (130, 257, 3, 2, True)])
def test_fused_svgp_elbo_shard(gpu, m, rows, d, P, q_diag):
    from gpflow_amd import ops
    rng = np.random.default_rng(9)
    X = rng.normal(size=(rows, d))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(rows, P))
    Z = X[:m] + 0.01 * rng.normal(size=(m, d)) if m <= rows else rng.normal(size=(m, d))
    q_mu = 0.1 * rng.normal(size=(m, P))
    if q_diag:
        q_sqrt = rng.uniform(0.3, 0.8, size=(m, P))
    else:
        q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m) for _ in range(P)])
    kw = dict(variance=1.1, lengthscales=np.sqrt(d) * (0.8 + 0.05 * np.arange(d)) if d > 1 else 0.7, noise_variance=0.1)
    out, info = ops.svgp_elbo_shard(_t(Z), _t(X), _t(Y), _t(q_mu), _t(q_sqrt), jitter=1e-6, **kw)
    ops.check_info(info)
    s_ref, kl_ref = orc.svgp_elbo_terms(X, Y, Z, q_mu, q_sqrt, whiten=True, **kw)
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[0], s_ref, rtol=1e-9)
    np.testing.assert_allclose(o[1], kl_ref, rtol=1e-12)


@pytest.mark.parametrize("m,rows,d,P", [(20, 50, 1, 2), (130, 257, 3, 2), (640, 1500, 8, 1), (1152, 300, 4, 3)])
def test_fused_svgp_elbo_shard_unwhitened_q_diag(gpu, m, rows, d, P):
    """gpk_svgp_elbo_shard(whiten = 0, q_diag = 1): [Kuu ; Kfu ; q_mu^T ; I] on one factorisation (the identity rows return Lm^-T)
    against the oracle's literal form -- gauss_kl's diag branch with K (kullback_leiblers.py:128-165) and the un-whitened
    conditional with a diagonal q_sqrt (conditionals/util.py:139-149).  With and without per-row noise variances."""
    from gpflow_amd import ops
    rng = np.random.default_rng(19)
    X = rng.normal(size=(rows, d))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(rows, P))
    Z = X[:m] + 0.05 * rng.normal(size=(m, d)) if m <= rows else rng.normal(size=(m, d))
    q_mu = 0.1 * rng.normal(size=(m, P))
    q_sqrt = rng.uniform(0.3, 0.8, size=(m, P))
    kw = dict(variance=1.1, lengthscales=np.sqrt(d) * (0.8 + 0.05 * np.arange(d)) if d > 1 else 0.7)
    for noise in (0.1, rng.uniform(0.05, 0.4, size=rows)):
        out, info = ops.svgp_elbo_shard(_t(Z), _t(X), _t(Y), _t(q_mu), _t(q_sqrt), jitter=1e-6, whiten=False,
                                        noise_variance=noise if np.isscalar(noise) else _t(noise), **kw)
        ops.check_info(info)
        s_ref, kl_ref = orc.svgp_elbo_terms(X, Y, Z, q_mu, q_sqrt, whiten=False, noise_variance=noise, **kw)
        o = out.cpu().numpy()
        # (kappa(Kuu) ~ 1e6 with the 1e-6 jitter: Kuu^-1-like terms carry kappa * eps)
        np.testing.assert_allclose(o[0], s_ref, rtol=1e-8)
        np.testing.assert_allclose(o[1], kl_ref, rtol=1e-8)


def test_per_row_noise_variances(gpu):
    """The (noise_variance, noise_rows) pair of the C-ABI: gpk_gaussian_varexp_sum, gpk_diag_add, gpk_gpr_lml and both fused ELBO
    drivers with one variance per data row (a heteroskedastic Gaussian likelihood, scalar_continuous.py:92-148)."""
    from gpflow_amd import ops
    rng = np.random.default_rng(23)
    rows, m, P, d = 333, 96, 3, 4
    Y = rng.normal(size=(rows, P)); fmean = rng.normal(size=(rows, P))
    s0 = rng.uniform(0, 0.5, size=rows); ssq = rng.uniform(0, 0.3, size=(P, rows))
    nv = rng.uniform(0.05, 2.0, size=rows)
    out, fvar = ops.gaussian_varexp_sum(_t(Y), _t(fmean), s0=_t(s0), ssq=_t(ssq), knn=[1.3], noise_variance=_t(nv), mean_const=0.1,
                                        want_fvar=True)
    fv = 1.3 - s0[:, None] + ssq.T
    np.testing.assert_allclose(out.cpu().numpy()[0], orc.gaussian_variational_expectations(fmean + 0.1, fv, Y, nv).sum(), rtol=1e-13)
    # a constant passed as a vector gives the scalar path's value (different rounding of log(nv): 1e-14)
    o1, _ = ops.gaussian_varexp_sum(_t(Y), _t(fmean), s0=_t(s0), ssq=_t(ssq), knn=[1.3], noise_variance=0.2)
    o2, _ = ops.gaussian_varexp_sum(_t(Y), _t(fmean), s0=_t(s0), ssq=_t(ssq), knn=[1.3], noise_variance=_t(np.full(rows, 0.2)))
    np.testing.assert_allclose(o2.cpu().numpy(), o1.cpu().numpy(), rtol=1e-13)
    with pytest.raises(ValueError):
        ops.gaussian_varexp_sum(_t(Y), _t(fmean), s0=None, ssq=None, knn=[1.3], noise_variance=_t(nv[:5]))
    A = rng.normal(size=(m, m + 3))
    A2 = ops.diag_add_(_t(A)[:, :m], _t(nv[:m])).cpu().numpy()
    np.testing.assert_array_equal(A2, A[:, :m] + np.diag(nv[:m]))
    # fused GPR LML and both fused ELBO drivers
    n = 520
    X = rng.normal(size=(n, d)); Yg = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(n, 2))
    nvn = rng.uniform(0.05, 0.5, size=n)
    kw = dict(variance=1.2, lengthscales=0.8 + 0.1 * np.arange(d))
    out, info = ops.gpr_lml(_t(X), _t(Yg), mean_const=0.05, noise_variance=_t(nvn), **kw)
    ops.check_info(info)
    np.testing.assert_allclose(out.cpu().numpy()[0], orc.gpr_log_marginal_likelihood(X, Yg, mean=0.05, noise_variance=nvn, **kw),
                               rtol=1e-10)
    Z = X[:m] + 0.05 * rng.normal(size=(m, d))
    q_mu = 0.1 * rng.normal(size=(m, 2))
    q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m) for _ in range(2)])
    for wh in (True, False):
        o, info = ops.svgp_elbo_shard(_t(Z), _t(X), _t(Yg), _t(q_mu), _t(q_sqrt), jitter=1e-6, whiten=wh, noise_variance=_t(nvn), **kw)
        ops.check_info(info)
        s_ref, kl_ref = orc.svgp_elbo_terms(X, Yg, Z, q_mu, q_sqrt, whiten=wh, noise_variance=nvn, **kw)
        np.testing.assert_allclose(o.cpu().numpy(), [s_ref, kl_ref], rtol=1e-8)
    o, info = ops.svgp_elbo_shard_sep(_t(Z), _t(X), _t(Yg), _t(q_mu), _t(q_sqrt), variances=[1.2, 0.9], lengthscales=[0.8, 1.1],
                                      families=["SquaredExponential", "SquaredExponential"], noise_variance=_t(nvn), jitter=1e-6)
    ops.check_info(info)
    ref = orc.svgp_elbo_separate(X, Yg, [Z, Z], q_mu, q_sqrt, variances=[1.2, 0.9], lengthscales_list=[0.8, 1.1], noise_variance=nvn)
    o = o.cpu().numpy()
    np.testing.assert_allclose(o[0] - o[1], ref, rtol=1e-8)


@pytest.mark.parametrize("m,rows,d,P", [(256, 300, 3, 1), (640, 1000, 8, 2), (1024, 2500, 8, 1), (1152, 777, 4, 3)])
def test_svgp_elbo_shard_column_groups(gpu, m, rows, d, P):
    """The fused shard over the column-group shapes of the extra-row solve / q_sqrt projection, against the oracle: one
    group (m = 256), ragged last group (640 = 512 + 128), shrinking tail groups (1024, 1152), several latents, ragged
    row counts; bulk GEMMs ticketed under the software CU reservation.  (With the A/B library, GPK_LIBRARY=libgpk_exp.so,
    the same test covers GPK_STREAM_PROJ=0/1 and GPK_SOFT_RESERVE=0/1 from the environment: tools/ab.sh.)"""
    from gpflow_amd import ops
    rng = np.random.default_rng(13)
    X = rng.normal(size=(rows, d))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(rows, P))
    Z = rng.normal(size=(m, d))
    q_mu = 0.1 * rng.normal(size=(m, P))
    q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m) for _ in range(P)])
    kw = dict(variance=1.1, lengthscales=np.sqrt(d) * (0.8 + 0.05 * np.arange(d)), noise_variance=0.1)
    out, info = ops.svgp_elbo_shard(_t(Z), _t(X), _t(Y), _t(q_mu), _t(q_sqrt), jitter=1e-6, **kw)
    ops.check_info(info)
    s_ref, kl_ref = orc.svgp_elbo_terms(X, Y, Z, q_mu, q_sqrt, whiten=True, **kw)
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[0], s_ref, rtol=1e-9)
    np.testing.assert_allclose(o[1], kl_ref, rtol=1e-12)
    out2, _ = ops.svgp_elbo_shard(_t(Z), _t(X), _t(Y), _t(q_mu), _t(q_sqrt), jitter=1e-6, **kw)
    np.testing.assert_array_equal(out2.cpu().numpy(), o)  # deterministic


@pytest.mark.parametrize("m,n,k,b_tri,c_lower", [(1536, 1280, 272, 0, False), (1408, 1408, 512, 0, True),
                                                  (1200, 1024, 1024, 1, False), (1100, 896, 896, 2, False)])
def test_gemm_nt_fast_path(gpu, m, n, k, b_tri, c_lower):
    """Shapes that select the software-pipelined 128x128x16 kernel (>= 24 tiles, K % 16 == 0, aligned rows): ragged
    row/column edges, beta != 0 (accumulators preloaded with C), lower-only tile enumeration, triangular K ranges."""
    from gpflow_amd import ops
    rng = np.random.default_rng(12)
    A = rng.normal(size=(m, k))
    B = rng.normal(size=(n, k))
    if b_tri == 1:
        B = np.triu(B)   # B[j, kk] == 0 for kk < j
    if b_tri == 2:
        B = np.tril(B)   # B[j, kk] == 0 for kk > j
    C = rng.normal(size=(m, n))
    out = ops.gemm_nt(_t(A), _t(B), alpha=-0.9, beta=1.1, C=_t(C), b_tri=b_tri, c_lower=c_lower).cpu().numpy()
    ref = -0.9 * A @ B.T + 1.1 * C
    if c_lower:
        np.testing.assert_allclose(np.tril(out), np.tril(ref), rtol=0, atol=2e-11)
    else:
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-11)


@pytest.mark.parametrize("np_,m,n,lower", [(1, 300, 300, True), (8, 2048, 2048, True), (5, 130, 17, False), (3, 257, 511, False),
                                          (4, 515, 515, True)])
def test_combine_parts(gpu, np_, m, n, lower):
    """gpk_combine_parts: alpha * sum of the split-K partial products in a fixed order; lower: tril with exact zeros and
    a scaled diagonal, entries above the diagonal never read (NaN there on purpose).  Bit-exact against the same
    left-to-right sum in NumPy."""
    import torch
    from gpflow_amd import ops
    rng = np.random.default_rng(9)
    parts = rng.normal(size=(np_, m, n))
    ref = np.zeros((m, n))
    for p in range(np_):
        ref = ref + parts[p]
    ref = -1.5 * ref
    src = parts.copy()
    if lower:
        src[:, np.triu(np.ones((m, n), dtype=bool), 1)] = np.nan
        ref = np.tril(ref)
        ref[np.diag_indices(min(m, n))] *= 0.5
    out = ops.combine_parts(_t(src), alpha=-1.5, lower=lower, diag_scale=0.5 if lower else 1.0).cpu().numpy()
    assert np.array_equal(out, ref)
    # strided view (a column block of a wider matrix) and a caller-provided output
    if not lower and n > 20:
        wide = _t(np.concatenate([parts, parts], axis=2))
        o = torch.empty((m, n - 3), dtype=torch.float64, device=wide.device)
        ops.combine_parts(wide[:, :, 3:n], out=o)
        r2 = np.zeros((m, n - 3))
        for p in range(np_):
            r2 = r2 + parts[p][:, 3:]
        assert np.array_equal(o.cpu().numpy(), r2)


@pytest.mark.parametrize("n", [300, 1024, 2048 + 40])
@pytest.mark.parametrize("a_tri,b_tri,c_lower", [(1, 1, True), (1, 1, False), (2, 1, False), (2, 2, False), (1, 2, False), (2, 0, False)])
def test_gemm_nt_triangular_a_hint(gpu, n, a_tri, b_tri, c_lower):
    """gpk_gemm_nt with the A-structure hint (b_tri bits 4-5): same result as the dense product of the same operands --
    the hint only shortens K ranges -- on the tiled kernel (aligned sizes), the generic one (ragged) and with c_lower."""
    from gpflow_amd import ops
    rng = np.random.default_rng(12)
    A = rng.normal(size=(n, n)); B = rng.normal(size=(n, n))
    A = np.triu(A) if a_tri == 1 else np.tril(A)
    if b_tri:
        B = np.triu(B) if b_tri == 1 else np.tril(B)
    out = ops.gemm_nt(_t(A), _t(B), alpha=0.75, b_tri=b_tri, c_lower=c_lower, a_tri=a_tri).cpu().numpy()
    ref = 0.75 * (A @ B.T)
    if c_lower:
        out, ref = np.tril(out), np.tril(ref)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("n,chunks", [(2048, 4), (1024, 4), (1536, 2), (2048, 8)])
@pytest.mark.parametrize("a_tri,b_tri", [(1, 1), (2, 1), (1, 2), (2, 2), (0, 1)])
def test_gemm_nt_k_split_of_a_triangular_product(gpu, n, chunks, a_tri, b_tri):
    """gpk_gemm_nt, b_tri bit 8: the batch entries are K chunks of one triangular x triangular product (strided views of the
    operands), the structure hints refer to the unsplit column index; the sum of the partial products is the product."""
    import torch
    from gpflow_amd import ops
    rng = np.random.default_rng(13)
    A = rng.normal(size=(n, n)); B = rng.normal(size=(n, n))
    if a_tri:
        A = np.triu(A) if a_tri == 1 else np.tril(A)
    B = np.triu(B) if b_tri == 1 else np.tril(B)
    At, Bt = _t(A), _t(B)
    kc = n // chunks
    A3 = torch.as_strided(At, (chunks, n, kc), (kc, n, 1))
    B3 = torch.as_strided(Bt, (chunks, n, kc), (kc, n, 1))
    parts = ops.gemm_nt(A3, B3, alpha=0.75, b_tri=b_tri, a_tri=a_tri, k_split=True)
    ref = 0.75 * (A @ B.T)
    for z in range(chunks):   # every partial product is the product of its chunk (no tile left unwritten)
        np.testing.assert_allclose(parts[z].cpu().numpy(), 0.75 * (A[:, z * kc:(z + 1) * kc] @ B[:, z * kc:(z + 1) * kc].T), rtol=0,
                                   atol=1e-10 * max(1.0, np.abs(ref).max()))
    out = ops.combine_parts(parts).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()))


def test_two_host_threads_one_device(gpu):
    """include/gpk.h, "Internal state and threading": factorisations with n > 128 share per-device streams and events,
    calls from several host threads are serialised by a per-device mutex while they ENQUEUE (the work itself overlaps on
    the device as far as its streams allow).  Two threads, each on its own stream, 12 factorisations each, against the
    results of the same calls made one after the other: bit for bit."""
    import threading
    import torch
    from gpflow_amd import ops
    rng = np.random.default_rng(77)
    probs = []
    for n, extra in ((700, 300), (1300, 40)):
        _, K = _spd(rng, n, noise=0.2)
        probs.append((n, _t(np.vstack([K, rng.normal(size=(extra, n))]))))
    expected = []
    for n, T0 in probs:
        T = T0.clone()
        _, info = ops.potrf_(T, n, zero_upper=True)
        ops.check_info(info)
        expected.append(T.cpu().numpy())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in probs]
    out, errs = [[] for _ in probs], []

    def work(i):
        try:
            n, T0 = probs[i]
            with torch.cuda.stream(streams[i]):
                for _ in range(12):
                    T = T0.clone()
                    _, info = ops.potrf_(T, n, zero_upper=True)
                    out[i].append((T, info))
                streams[i].synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream())
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(probs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(len(probs)):
        assert len(out[i]) == 12
        for T, info in out[i]:
            assert int(info.cpu()[0]) == 0
            assert np.array_equal(T.cpu().numpy(), expected[i])


@pytest.mark.parametrize("m,n,k,view", [(1500, 1301, 256, 0), (1300, 1290, 320, 0), (1290, 1408, 256, 1), (1409, 1282, 272, 2)])
def test_gemm_nt_fast_path_c_access(gpu, m, n, k, view):
    """The 128x128x16 kernel preloads / stores its accumulators with 16-byte accesses when the tile's columns lie inside
    the matrix and C's rows are 16-byte aligned, element by element otherwise: odd n (odd leading dimension), even n that
    is no multiple of 128 (only the last column tile falls back), C as a view that starts 8 bytes into a row (view 1) and
    a view with an odd leading dimension (view 2).  alpha, beta != 0, 1 so that both the preload and the store count."""
    import torch
    from gpflow_amd import ops
    rng = np.random.default_rng(77)
    A, B, C = rng.normal(size=(m, k)), rng.normal(size=(n, k)), rng.normal(size=(m, n))
    if view == 0:
        Cd = _t(C)
    else:
        big = torch.zeros((m, n + (2 if view == 1 else 3)), dtype=torch.float64, device=_t(A).device)
        Cd = big[:, 1:1 + n]
        Cd.copy_(_t(C))
    out = ops.gemm_nt(_t(A), _t(B), alpha=0.7, beta=-1.3, C=Cd).cpu().numpy()
    np.testing.assert_allclose(out, 0.7 * A @ B.T - 1.3 * C, rtol=0, atol=2e-11)
    if view:
        bo = big.cpu().numpy()
        assert np.all(bo[:, 0] == 0) and np.all(bo[:, 1 + n:] == 0)   # nothing written outside the view


def test_host_mailbox(gpu):
    """gpk_publish_host: device scalars + status land in pinned host memory, sequence word last; successive posts."""
    import torch
    from gpflow_amd import ops
    box = ops.HostMailbox(3)
    dev = ops.device()
    for i in range(5):
        src = torch.tensor([1.5 + i, -2.0, 3.25e10], dtype=torch.float64, device=dev)
        info = torch.tensor([i], dtype=torch.int32, device=dev)
        box.post(src, info)
        vals, status = box.wait()
        np.testing.assert_array_equal(vals, [1.5 + i, -2.0, 3.25e10])
        assert status == i
    box.post(torch.ones(3, dtype=torch.float64, device=dev))
    vals, status = box.wait()
    assert status == 0 and np.all(vals == 1.0)


def test_chain_handoff_mode_and_serialised_kernels(gpu):
    """The factorisation's chain hands over through flag words written and awaited by kernels only -- entry signals, gate / store
    kernels, bounded in-kernel polls (gpk_chain_handoff_mode() == 2, include/gpk.h) -- when kernels of two streams really run
    concurrently, and falls back to events when a tool serialises kernels: there the polls could never be satisfied (seen with
    rocprofv3 --pmc).  AMD_SERIALIZE_KERNEL=3 makes the HIP
    runtime wait around every launch: a child process under it must finish, report mode 0 and the same factor."""
    import subprocess
    import sys
    from gpflow_amd import _lib, ops
    rng = np.random.default_rng(5)
    _, K = _spd(rng, 1100)
    T = _t(np.vstack([K, rng.normal(size=(300, 1100))]))
    _, info = ops.potrf_(T, 1100, zero_upper=True)
    ops.check_info(info)
    assert _lib.load().gpk_chain_handoff_mode() == 2
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_gpu_primitives as tp\n"
        "from gpflow_amd import _lib, ops\n"
        "rng = np.random.default_rng(5)\n"
        "_, K = tp._spd(rng, 1100)\n"
        "T = tp._t(np.vstack([K, rng.normal(size=(300, 1100))]))\n"
        "_, info = ops.potrf_(T, 1100, zero_upper=True)\n"
        "ops.check_info(info)\n"
        "L = T.cpu().numpy()[:1100]\n"
        "print('MODE', _lib.load().gpk_chain_handoff_mode(), 'ERR', float(np.abs(L @ L.T - K).max()))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMD_SERIALIZE_KERNEL="3")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("MODE")][0].split()
    assert int(line[1]) == 0 and float(line[3]) < 5e-12, line


# ---- reverse-pass glue as single launches (include/gpk.h: gpk_moment_rows, gpk_stationary_adjoint_tail, gpk_adam_step,
# gpk_symmetrize).  The reference has no counterpart (TF autodiff / tf.optimizers.Adam): the checker is the plain
# NumPy arithmetic each header comment states.
@pytest.mark.parametrize("n2,d", [(1, 1), (300, 3), (8192, 8), (1000, 16)])
def test_moment_rows(gpu, n2, d):
    from gpflow_amd import ops
    rng = np.random.default_rng(1)
    B = rng.normal(size=(n2, d))
    Vt = ops.moment_rows(_t(B)).cpu().numpy()
    ref = np.vstack([np.ones((1, n2)), B.T, (B * B).T])
    np.testing.assert_array_equal(Vt, ref)


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("n1,d,with_sum", [(1, 1, False), (77, 3, True), (2048, 8, False), (5000, 16, False), (300, 100, True)])
def test_stationary_adjoint_tail(gpu, symmetric, n1, d, with_sum):
    from gpflow_amd import ops
    rng = np.random.default_rng(2)
    R, A = rng.normal(size=(n1, 1 + 2 * d)), rng.normal(size=(n1, d))
    ls = 0.6 + 0.05 * np.arange(d)
    sk = np.array([3.25]) if with_sum else None
    dvar, dls, Abar = ops.stationary_adjoint_tail(_t(R), _t(A), _t(ls), variance=1.7, symmetric=symmetric,
                                                  sum_kbar_k=None if sk is None else _t(sk))
    rs, GB, GB2 = R[:, :1], R[:, 1:1 + d], R[:, 1 + d:]
    T = GB - A * rs
    if symmetric:
        Ab = 2.0 * T / ls ** 2
        dl = -(A * Ab).sum(0) / ls
    else:
        Ab = T / ls ** 2
        dl = (GB2 - A * (GB + T)).sum(0) / ls ** 3
    dv = (rs.sum() if sk is None else sk[0]) / 1.7
    np.testing.assert_allclose(Abar.cpu().numpy(), Ab, rtol=1e-14, atol=1e-14)
    scale = np.abs(R).sum() + 1.0   # (sums of n1 terms of mixed sign: tolerance relative to the magnitude summed)
    np.testing.assert_allclose(dls.cpu().numpy(), dl, rtol=0, atol=1e-13 * scale / ls.min() ** 3)
    np.testing.assert_allclose(dvar.cpu().numpy(), [dv], rtol=0, atol=1e-13 * scale)


@pytest.mark.parametrize("n,maximise", [(1, False), (1000, True), (2048 * 2048 + 3, False)])
def test_adam_step(gpu, n, maximise):
    import torch
    from gpflow_amd import ops
    rng = np.random.default_rng(3)
    p, g, m, v = rng.normal(size=n), rng.normal(size=n), 0.1 * rng.normal(size=n), np.abs(rng.normal(size=n))
    P, G, Mm, V = _t(p), _t(g), _t(m), _t(v)
    ops.adam_step_(P, G, Mm, V, beta1=0.9, beta2=0.999, epsilon=1e-7, step=2.5e-3, maximise=maximise)
    gg = -g if maximise else g
    m2 = 0.9 * m + (1.0 - 0.9) * gg
    v2 = 0.999 * v + (1.0 - 0.999) * gg * gg
    p2 = p - 2.5e-3 * m2 / (np.sqrt(v2) + 1e-7)
    # (fused multiply-adds on the device: a few ulp of the terms, which cancel in m)
    np.testing.assert_allclose(Mm.cpu().numpy(), m2, rtol=0, atol=1e-15)
    np.testing.assert_allclose(V.cpu().numpy(), v2, rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(P.cpu().numpy(), p2, rtol=1e-13, atol=1e-14)
    assert torch.equal(G.cpu(), torch.from_numpy(g))


@pytest.mark.parametrize("n", [1, 31, 32, 100, 2048])
def test_symmetrize(gpu, n):
    from gpflow_amd import ops
    rng = np.random.default_rng(4)
    S = rng.normal(size=(n, n))
    out = ops.symmetrize_(_t(S)).cpu().numpy()
    np.testing.assert_array_equal(out, 0.5 * (S + S.T))
    # a strided view (row stride > n): only the n x n block is touched
    big = rng.normal(size=(n, n + 5))
    Tb = _t(big)
    ops.symmetrize_(Tb[:, :n])
    res = Tb.cpu().numpy()
    np.testing.assert_array_equal(res[:, :n], 0.5 * (big[:, :n] + big[:, :n].T))
    np.testing.assert_array_equal(res[:, n:], big[:, n:])


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (300, 77, 3), (8192, 2048, 1), (513, 1001, 16)])
def test_lowrank_axpy(gpu, m, n, k):
    from gpflow_amd import ops
    rng = np.random.default_rng(6)
    X, U, V = rng.normal(size=(m, n)), rng.normal(size=(m, k)), rng.normal(size=(n, k))
    out = ops.lowrank_axpy(-0.7, _t(X), _t(U), _t(V)).cpu().numpy()
    np.testing.assert_allclose(out, -0.7 * X + U @ V.T, rtol=0, atol=1e-14 * k)
