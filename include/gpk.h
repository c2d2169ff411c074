/* gpk.h -- C-ABI of libgpk.so: the MI355X (gfx950) dense-GP hot path behind gpflow_amd.
 *
 * The reference (GPflow 2.9.2) has no C/FFI boundary: every FLOP of this path is a TensorFlow op
 * called from Python.  Each entry point below replaces one TF-op call site (or a fused group of
 * them); the reference file:line it stands in for is cited per function (paths relative to the
 * GPflow tree).  The reference-side binding a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *   - every matrix pointer is a DEVICE pointer to row-major fp64 (gpflow default_float,
 *     config/__config__.py:99); leading dimensions are in elements;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work,
 *     they never synchronise; scratch comes from the caller's workspace;
 *   - small hyper-parameter vectors (lengthscales) are HOST pointers, copied into kernel arguments;
 *   - return value: 0 ok, <0 bad argument (GPK_E_*), >0 HIP runtime error code (hipError_t);
 *   - numerical failure (non-positive pivot) is reported LAPACK-style through a device int
 *     `info` (0 = ok, j+1 = first bad pivot column) that the caller reads when it next syncs
 *     (TF raises InvalidArgumentError "Cholesky decomposition was not successful" at the same spot);
 *     INT_MAX = an internal stream hand-off of the factorisation timed out (0.5 s; never observed -- the bounded wait exists so
 *     that a lost hand-off cannot hang the device);
 *
 * Internal state and threading (the complete list; nothing else in the library is mutable)
 *   - gpk_potrf with n > 128 (and the two fused drivers, which call it) uses per-device state created lazily on the
 *     first such call: five internal HIP streams (P panel, high priority / X side / one placeholder that fixes the
 *     stream-to-hardware-queue layout / Bs bulk-small / B bulk, CU-masked) and a pool of timing-disabled events that grows
 *     to 2 * panels + 8, and 4 KB of device memory for hand-off words: single-leaf panels of the latency chain are handed
 *     over between these streams with hipStreamWaitValue32 / hipStreamWriteValue32 and an in-kernel poll instead of event
 *     packets (an event record / wait between two kernels of one stream costs 4.6 / 6.3 us on MI355X; the poll is bounded:
 *     after 0.5 s a waiting kernel proceeds rather than hang the device).  That first call also runs a ~1 ms self-check of the stream layout (a few hundred empty kernels;
 *     the ONLY place the library synchronises) and creates the streams again if a pair of them hands kernels over slowly
 *     (gpk_stream_selfcheck).  Afterwards no device memory is allocated and nothing synchronises: work is forked from and
 *     joined to the caller's stream with events only.  (The streams are created in an order that keeps the panel and bulk
 *     streams on different microengine pipes -- INTEGRATION.md section 4: create this state, i.e. make one such call,
 *     before other libraries of the process create their streams.)
 *   - One recursive mutex per device serialises the ENQUEUE section of these calls, so they may be issued from any
 *     number of host threads and on any caller streams; factorisations of one device share the internal streams and
 *     therefore execute one after the other on the GPU.
 *   - Every other entry point is stateless and reentrant (kernel attributes are set through thread-safe function-local
 *     statics; the GEMM launcher keeps the id of the kernel it picked last in a thread_local for the profiling facility).
 *     The library never reads the environment.
 *   - gpk_profile_gemm_* is a measurement facility for bench.py: process-global, not thread-safe, off by default.
 */
#ifndef GPK_H
#define GPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libgpk.so is built with -fvisibility=hidden: exactly the functions declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define GPK_NB 128 /* Cholesky inner block: diag-block inverses are GPK_NB x GPK_NB */
#define GPK_MAX_D 64 /* max input dimension of the covariance builder */

#define GPK_E_ARG (-1)
#define GPK_E_WORKSPACE (-2)
#define GPK_E_UNSUPPORTED (-3)

/* stationary kernel families (gpflow/kernels/stationaries.py) */
#define GPK_KERN_SE 0       /* SquaredExponential.K_r2 :209-210 */
#define GPK_KERN_MATERN12 1 /* :254-255 */
#define GPK_KERN_MATERN32 2 /* :281-283 */
#define GPK_KERN_MATERN52 3 /* :311-313 */

const char* gpk_version(void);

/* ---- covariance builder ---------------------------------------------------------------------
 * K[i,j] = variance * k(r2(X1_i / ls, X2_j / ls)) (+ diag_add if i == j and X2 == NULL)
 * Replaces square_distance (utilities/ops.py:105-122, the expansion formula, kept), Stationary.scale
 * (stationaries.py:77-79), K_r2 (:209-210 ...), add_noise_cov (utilities/model_utils.py:33-38) and
 * the jitter add of Kuu (covariances/kuus.py:33).
 *   X1 [n1,d] ldx1; X2 [n2,d] ldx2 or NULL (symmetric: X2 = X1);  K [n1,n2] ldk
 *   ls: HOST pointer, d entries if ard else 1;  lower_only: with X2 == NULL write only tiles that
 *   touch the lower triangle (what the Cholesky reads). */
int gpk_kernel_matrix(void* stream, int family, const double* X1, int n1, long ldx1,
                      const double* X2, int n2, long ldx2, int d, const double* ls_host, int ard,
                      double variance, double diag_add, int lower_only, double* K, long ldk);

/* A[i,i] += v[i], i < n (v: DEVICE pointer):  add_noise_cov / add_likelihood_noise_cov with a per-row likelihood variance
 * (utilities/model_utils.py:33-38, 46-50) -- the heteroskedastic counterpart of gpk_kernel_matrix's scalar diag_add. */
int gpk_diag_add(void* stream, double* A, int n, long lda, const double* v);

/* out = G .* k(X1, X2) (op 1), G + k(X1, X2) (op 2) or G .* (-2 dk/dr2)(X1, X2) (op 3), k recomputed from the inputs (one read of G, one write; G may
 * alias out).  Replaces the tf.multiply / tf.add_n reductions of Product / Sum kernels (kernels/base.py:216-220,
 * 283-329): the second factor / term is folded into the first matrix in place instead of being materialised.
 * X2 == NULL: K(X1, X1), and diag_add goes onto the diagonal of the COMBINED result (noise / jitter).
 * op 3 is the reverse pass of the Matern family (stationaries.py:254-313: r = sqrt(max(r2, 1e-36)), K_r): with
 * r2 the SCALED squared distance, every lengthscale / input gradient contracts Kbar .* dk/dr2 with d r2 / d theta;
 * -2 dk/dr2 equals k for the SquaredExponential, so the same contraction code serves all four families.  X2 == NULL
 * writes exact zeros on the diagonal (r2_ii = 0 identically: no gradient flows through it). */
int gpk_kernel_matrix_combine(void* stream, int family, int op, const double* X1, int n1, long ldx1,
                              const double* X2, int n2, long ldx2, int d, const double* ls_host, int ard,
                              double variance, double diag_add, const double* G, long ldg, double* out, long ldo);

/* = gpk_kernel_matrix_combine(op 1) with X2 given: the elementwise factor of every kernel-parameter gradient,
 * dF/dtheta = sum_ij Kbar_ij dK_ij/dtheta with dK/dtheta = K .* (...) for the stationary kernels -- the reverse pass
 * of Stationary.K (stationaries.py:103-116, 209-210) that TF autodiff builds for optimizers/scipy.py:322-331
 * (SURVEY 8f row 1). */
int gpk_kernel_matrix_hadamard(void* stream, int family, const double* X1, int n1, long ldx1,
                               const double* X2, int n2, long ldx2, int d, const double* ls_host, int ard,
                               double variance, const double* G, long ldg, double* out, long ldo);

/* ---- trapezoidal Cholesky ---------------------------------------------------------------------
 * A is [(n + extra) x n] row-major.  Top n x n block (lower triangle read): K -> L in place,
 * K = L L^T (tf.linalg.cholesky call sites: gpr.py:102, posteriors.py:422,703, conditionals/util.py:67,
 * kullback_leiblers.py:107).  The `extra` rows below it hold right-hand sides B [extra, n] and come
 * back as B L^-T, i.e. (L^-1 B^T)^T -- tf.linalg.triangular_solve(L, B^T, lower=True)
 * (conditionals/util.py:125, logdensities.py:150, kullback_leiblers.py:114,152) fused into the
 * factorisation's panel updates.  The strict upper triangle of the top block is left untouched
 * unless zero_upper != 0.  `batch` matrices at stride strideA (SeparateIndependent: the
 * tf.map_fn loop of conditionals/util.py:618 becomes one batched launch sequence).
 * invd: [batch, ceil(n/NB), NB, NB] output, the inverses of L's diagonal blocks (reused by gpk_trsm);
 * gpk_invd_elems() gives its size in doubles.  info: device int[batch] (may be NULL). */
size_t gpk_invd_elems(int n, int batch);
int gpk_potrf(void* stream, double* A, int n, int extra, long lda, int batch, long strideA,
              double* invd, int zero_upper, int* info);

/* Result of the init-time stream-layout self-check (see "Internal state and threading"): microseconds per cross-stream
 * kernel hand-off between the library's panel / side / bulk-small streams, as measured now (us_now[3]) and on the first
 * stream set (us_first[3]); *recreated = 1 if the first set was slow (> 30 us: two active hardware queues on one
 * microengine pipe) and the streams were created again.  GPK_E_UNSUPPORTED before the first factorisation with n > 128. */
int gpk_stream_selfcheck(double* us_now, double* us_first, int* recreated);
/* How the factorisation's latency chain hands over between the internal streams on the current device:
 *    2 = flag words in device memory, written and awaited by KERNELS only -- the entry signal of a GEMM kernel, a one-thread store
 *        kernel, a one-wave gate kernel, the bounded in-kernel poll of the strip -- with no queue packet between the chain's
 *        kernels (the product library, round 6);
 *    1 = the same words written / awaited with hipStreamWriteValue32 / hipStreamWaitValue32 (rounds 5; the A/B build with
 *        GPK_GATE_KERNELS=0).  Observed on ROCm 7.2.0 / gfx950: on a CU-MASKED stream such a write overtook the kernel queued
 *        before it (a wrong factor at n = 5000) -- stream memory operations were therefore only ever used on plain streams, and the
 *        product no longer uses them at all: kernels of one stream execute in order by definition;
 *    0 = events -- when the first factorisation finds that kernels of two streams do NOT run at the same time (a tool that
 *        serialises kernels, e.g. rocprofv3 --pmc, would deadlock the polls);
 *   -1 = no factorisation with n > 128 has been issued on this device yet.
 * Every wait is bounded (0.5 s); one that expires sets the status word to INT_MAX (see "info") and the call returns. */
int gpk_chain_handoff_mode(void);

/* gpk_potrf for callers that also need the explicit inverse factor (the reverse pass: gradients.py; the reference
 * gets the same quantities from the triangular solves inside TF's Cholesky gradient).  A is [n + extra + n, lda]:
 * the square block, `extra` right-hand-side rows, and n more rows that the CALL overwrites with the identity and
 * returns as L^-T (upper triangular, exact zeros below the diagonal).  Because row j of that block stays zero left
 * of column j until its column group is reached, the identity rows cost n^3/3 flop, not the n^3 of n dense rows.
 * batch 1.  Everything else as gpk_potrf. */
int gpk_potrf_inv(void* stream, double* A, int n, int extra, long lda, double* invd, int zero_upper, int* info);

/* inverses of the diagonal NB-blocks of an existing lower factor L [n,n] (for gpk_trsm on a cached
 * L: GPRPosterior cache (err, Lm), posteriors.py:415-432). invd [batch, ceil(n/NB), NB, NB]. */
int gpk_trtri_blocks(void* stream, const double* L, int n, long ldl, int batch, long strideL,
                     double* invd);

/* ---- triangular solve against an existing factor (right-side, row-major form) -------------------
 * trans = 0:  B <- B L^-T   (rows of B are right-hand sides;  = (L^-1 B^T)^T, util.py:125)
 *             pass L (lower, ldl) and invd.
 * trans = 1:  B <- B L^-1   (= (L^-T B^T)^T, triangular_solve(adjoint(Lm), A, lower=False) util.py:139)
 *             pass LT = L^T (upper, row-major) as `L` and the blockwise transposed inverses invdT as
 *             `invd`; both come from gpk_transpose_factor.
 * B [m,n] ldb, solved in place. */
int gpk_trsm(void* stream, int trans, const double* L, long ldl, const double* invd, int n,
             double* B, int m, long ldb, int batch, long strideL, long strideB);
int gpk_transpose_factor(void* stream, const double* L, long ldl, const double* invd, int n,
                         double* LT, long ldlt, double* invdT);

/* ---- GEMM  C = alpha * A * B^T + beta * C  (A [m,k], B [n,k], C [m,n]; fp64 MFMA) ----------------
 * tf.linalg.matmul call sites (conditionals/util.py:129,144,157,162; posteriors.py:734,803-818).
 * b_tri, bits 0-1: 0 dense; 1 B is upper in [n,k] (B[j,kk] == 0 for kk < j, must be stored as zeros);
 *        2 B is lower (B[j,kk] == 0 for kk > j).  Bits 4-5 (optional hint about A, m <= k): 16 = A is upper
 *        (A[i,kk] == 0 for kk < i), 32 = A is lower (A[i,kk] == 0 for kk > i), stored as zeros as well; a tile's K range
 *        is then the intersection of both structures -- the triangular x triangular products of the reverse pass
 *        (K^-1 = L^-T L^-1, the Cholesky adjoint) at half the multiply-adds.  c_lower: only tiles touching the lower
 *        triangle.  Bit 8 (256, round 6): the `batch` entries are consecutive K chunks of ONE triangular product -- entry z
 *        holds columns z k .. (z + 1) k of both operands (strideA = strideB = k on the unsplit arrays), the triangular
 *        statements refer to the unsplit column index, k must be a multiple of 16 -- and the caller sums the partial
 *        products (gpk_combine_parts).  A 2048^3 triangular x triangular product is 256 output tiles whose longest walks
 *        K = 2048 (245 us); as 4 chunks it is one round of 480 non-empty tiles (gradients.py, cholesky_adjoint). */
int gpk_gemm_nt(void* stream, int m, int n, int k, double alpha, const double* A, long lda,
                const double* B, long ldb, double beta, double* C, long ldc, int b_tri,
                int c_lower, int batch, long strideA, long strideB, long strideC);

/* out[j,i] = in[i,j]; mode 0 plain, 1 keep the lower triangle of `in` only (band_part(q_sqrt,-1,0),
 * conditionals/util.py:151, kullback_leiblers.py:120), 2 keep upper only. in [rows,cols]. */
int gpk_transpose(void* stream, const double* in, int rows, int cols, long ldin, double* out,
                  long ldout, int mode, int batch, long stride_in, long stride_out);

/* Row statistics of At [rows, m] (= A^T of the conditional), one pass over At:
 *   sumsq[b]  = beta*sumsq[b] + alpha * sum_k At[b,k]^2        (util.py:133 reduce_sum(square(A), -2))
 *   mv[b,p]   = sum_k At[b,k] V[k,p]                           (util.py:144  A^T f;  V [m,P])
 *   wsq[p,b]  = sum_k (At[b,k] W[k,p])^2                       (util.py:149,164 q_diag; W [m,P])
 * any of sumsq / (V,mv) / (W,wsq) may be NULL. */
int gpk_row_stats(void* stream, const double* At, int rows, int m, long ldat, const double* V,
                  const double* W, int P, double alpha, double beta, double* sumsq, double* mv,
                  double* wsq);
int gpk_row_sumsq(void* stream, const double* A, int rows, int cols, long lda, double alpha,
                  double beta, double* out);
/* out[i] = beta*out[i] + alpha * sum_j A[i,j] B[i,j]   (tf.reduce_sum(Kuf * matmul(Qinv, Kuf), -2),
 * posteriors.py:818, in the row-major transposed form) */
int gpk_row_dot(void* stream, const double* A, long lda, const double* B, long ldb, int rows,
                int cols, double alpha, double beta, double* out);

/* Projection onto the variational square roots (util.py:151-164, triangular-aware, never
 * materialises LTA):  ssq[p,b] = sum_j ( sum_k At[b,k] Lq_p[k,j] )^2
 * LqT [P, m, ldl] = tril(q_sqrt_p)^T, built with gpk_transpose(mode 1). */
size_t gpk_project_workspace_bytes(int rows, int m, int P);
int gpk_project(void* stream, const double* At, int rows, int m, long ldat, const double* LqT,
                long ldl, int P, double* ssq, void* ws, size_t ws_bytes);
/* The same with one At PER latent, At_p = At + p * strideAt (strideAt = 0: gpk_project): the SeparateIndependent
 * conditional (util.py:566-629), whose tf.map_fn over the P problems becomes ONE launch over the batched trapezoid. */
int gpk_project_batched(void* stream, const double* At, int rows, int m, long ldat, long strideAt,
                        const double* LqT, long ldl, int P, double* ssq, void* ws, size_t ws_bytes);

/* ---- scalar tails (deterministic two-stage reductions) ---------------------------------------------
 * Gaussian variational expectations summed over rows and outputs
 * (likelihoods/scalar_continuous.py:139-148 then tf.reduce_sum, svgp.py:181):
 *   fvar[b,p] = knn - s0[b | p,b] + ssq[p,b]
 *   out[0] = sum_b sum_p -0.5 log 2pi - 0.5 log nv - 0.5 ((Y[b,p] - fmean[b,p] - mean_const)^2 + fvar) / nv
 * s0 [rows] or [P,rows] (s0_per_latent), may be NULL; ssq [P,rows] may be NULL; fvar_out [rows,P]
 * optional.  knn: HOST pointer, P values if knn_per_latent else 1.
 * noise_rows: DEVICE pointer to one noise variance PER ROW [rows] -- a heteroskedastic Gaussian likelihood,
 * Gaussian(variance=Function | scale=Function) evaluated at the rows (scalar_continuous.py:92-111, 139-148: nv becomes
 * nv[b]) -- or NULL for the constant `noise_variance`.  The same pair (noise_variance, noise_rows) appears in the fused
 * drivers below with the same meaning. */
size_t gpk_reduce_workspace_bytes(int n);
int gpk_gaussian_varexp_sum(void* stream, const double* Y, long ldy, const double* fmean, int rows,
                            int P, const double* s0, int s0_per_latent, const double* ssq,
                            const double* knn_host, int knn_per_latent, double noise_variance,
                            const double* noise_rows, double mean_const, double* fvar_out, double* out, void* ws,
                            size_t ws_bytes);

/* whitened KL (kullback_leiblers.py:98-165 with K None):
 *   out[0] = 0.5 * ( sum q_mu^2 - M*P - sum log diag(Lq)^2 + sum tril(Lq)^2 )
 * q_sqrt [P,m,m] (q_diag = 0) or [m,P] (q_diag = 1). */
int gpk_gauss_kl_white(void* stream, const double* q_mu, const double* q_sqrt, int m, int P,
                       int q_diag, double* out, void* ws, size_t ws_bytes);

/* out[b] = sum_i log(L_b[i,i]) (logdensities.py:154, kullback_leiblers.py:159-160) */
int gpk_sum_log_diag(void* stream, const double* L, int n, long ldl, int batch, long strideL,
                     double* out);
/* out[0] = sum of squares of A [rows,cols] (upper_only: c >= r) -- mahalanobis / trace terms */
int gpk_sumsq(void* stream, const double* A, int rows, int cols, long lda, int upper_only,
              double* out, void* ws, size_t ws_bytes);

/* out [m,n] = alpha * sum_p parts[p]  (parts [nparts, m, n], row stride ldp, part stride stride_part), summed in the
 * order p = 0, 1, ... (deterministic).  lower != 0: entries above the diagonal are written as zeros and never read,
 * the diagonal is multiplied by diag_scale -- i.e. tril(.) (tf.linalg.band_part(., -1, 0), conditionals/util.py:151)
 * or, with diag_scale = 0.5, the Phi(.) of the Cholesky adjoint.  The reduction step of the split-K products of the
 * reverse pass (gradients.py), where TF autodiff's matmul gradients need none. */
int gpk_combine_parts(void* stream, const double* parts, int nparts, long stride_part, int m, int n, long ldp,
                      double alpha, int lower, double diag_scale, double* out, long ldo);

/* ---- glue of the reverse pass as single launches (round 6) -------------------------------------------------
 * The reference gets all of this from TF autodiff and tf.optimizers.Adam (optimizers/scipy.py:322-331,
 * gps_for_big_data.pct.py:207-228); here the reverse pass is written out (gpflow_amd/gradients.py) and these entry
 * points replace ~70 elementwise launches of a few thousand elements each at the end of a training step.
 *
 * gpk_moment_rows:  Vt [1 + 2 d, n2] = [1; B^T; (B^T).^2] for B [n2, d] -- the right-hand side of G [1, B, B^2],
 *   the contraction that turns G = Kbar .* K into lengthscale / input gradients (stationaries.py:209-210 under autodiff). */
int gpk_moment_rows(void* stream, const double* B, long ldb, int n2, int d, double* Vt, long ldv);
/* gpk_stationary_adjoint_tail:  from R [n1, 1 + 2 d] = G [1, B, B^2] (columns: row sums, G B, G B^2), the first kernel
 *   argument A [n1, d] and the lengthscales ls_dev [d] (device):
 *     T = R[:, 1:1+d] - A .* R[:, 0]
 *     symmetric != 0 (B is A, Kbar symmetric):  Abar = 2 T / ls^2,  d/dls = -colsum(A .* Abar) / ls
 *     else:                                      Abar = T / ls^2,    d/dls = colsum(R[:, 1+d:] - A .* (R[:, 1:1+d] + T)) / ls^3
 *     d/dvariance = (sum_kbar_k ? sum_kbar_k[0] : sum(R[:, 0])) / variance
 *   Abar [n1, d]; small [1 + d] = (d/dvariance + dvar_add, d/dls).  accumulate != 0: both are ADDED to what Abar / small
 *   hold (the two adjoints of one kernel -- Kuf and Kuu -- leave their sum).  One workgroup, fixed summation order. */
int gpk_stationary_adjoint_tail(void* stream, const double* R, long ldr, const double* A, long lda, int n1, int d,
                                const double* ls_dev, double variance, int symmetric, const double* sum_kbar_k,
                                double* Abar, long ldab, double* small, int accumulate, double dvar_add);
/* gpk_adam_step:  tf.keras Adam on one variable of n doubles, in place:  g' = maximise ? -g : g,
 *   m = beta1 m + (1 - beta1) g',  v = beta2 v + (1 - beta2) g'^2,  p -= step m / (sqrt(v) + epsilon),
 *   step = lr sqrt(1 - beta2^t) / (1 - beta1^t) computed by the caller. */
int gpk_adam_step(void* stream, double* p, const double* g, double* m, double* v, long n, double beta1, double beta2,
                  double epsilon, double step, int maximise);
/* gpk_lowrank_axpy:  out [m, n] = alpha X + U V^T  for thin U [m, k], V [n, k], k <= 16 (out may be X): the first two terms of
 *   At_bar = r q_mu^T - 2 c P At + 2 c sum_p W_p Lq_p^T in one pass (the GEMM that adds the third has beta = 1). */
int gpk_lowrank_axpy(void* stream, double alpha, const double* X, long ldx, const double* U, long ldu, const double* V, long ldv,
                     int m, int n, int k, double* out, long ldo);
/* gpk_symmetrize:  S = (S + S^T) / 2 in place, S [n, n] -- the last step of the Cholesky adjoint. */
int gpk_symmetrize(void* stream, double* S, int n, long lds);

/* ---- fused drivers -----------------------------------------------------------------------------------
 * GPR.log_marginal_likelihood (gpr.py:91-107), stationary kernel, Gaussian noise, constant mean:
 * builds K(X,X)+noise*I (lower) with (Y-mean)^T as extra rows into ws, factors, reduces.
 *   out[0] = LML (summed over the P columns of Y); info: device int.
 * noise_rows (device, [n]) != NULL: K(X,X) + diag(noise_rows) instead -- add_likelihood_noise_cov with
 * likelihood.variance_at(X) (utilities/model_utils.py:46-50, gpr.py:100-101). */
size_t gpk_gpr_lml_workspace_bytes(int n, int d, int P);
int gpk_gpr_lml(void* stream, int family, const double* X, int n, int d, long ldx, const double* Y,
                int P, long ldy, const double* ls_host, int ard, double variance,
                double noise_variance, const double* noise_rows, double mean_const, double* out, int* info,
                void* ws, size_t ws_bytes);

/* One shard of SVGP.elbo (svgp.py:166-181), whitened, one kernel shared by all P latents
 * (IndependentPosteriorSingleOutput posteriors.py:828-841 and the SharedIndependent branch :849-861):
 *   out[0] = sum over this shard's rows of var_exp   (all-reduced across ranks by the caller)
 *   out[1] = KL (replicated; identical on every rank)
 * q_diag: q_sqrt is [m,P] instead of [P,m,m].
 * whiten = 0 (full q_sqrt): KL against N(0, Kuu) (kullback_leiblers.py:98-165 with K) and the un-whitened conditional
 * (conditionals/util.py:128-167, white = False) on ONE factorisation -- [Kuu ; Kfu ; q_mu^T ; tril(q_sqrt_p)^T] as one
 * trapezoid; the reference factors Kuu twice and solves the minibatch columns twice.  whiten = 0 with q_diag (round 5;
 * kullback_leiblers.py:128-165 diag branch, util.py:139-149): the trapezoid is [Kuu ; Kfu ; q_mu^T ; I], the identity rows
 * return Lm^-T, whose row norms are diag(Kuu^-1) (trace term) and which turns the second solve of the minibatch columns
 * into one triangular-K GEMM.  The workspace query takes the same (q_diag, whiten) pair as the call. */
size_t gpk_svgp_elbo_workspace_bytes(int m, int rows, int d, int P, int q_diag, int whiten);
int gpk_svgp_elbo_shard(void* stream, int family, const double* Z, int m, long ldz, const double* Xb,
                        const double* Yb, int rows, long ldxb, long ldyb, int d, int P,
                        const double* ls_host, int ard, double variance, double noise_variance,
                        const double* noise_rows, double jitter, double mean_const, const double* q_mu,
                        const double* q_sqrt, int q_diag, int whiten, double* out, int* info,
                        void* ws, size_t ws_bytes);

/* The same for SEPARATE kernels per latent (SeparateIndependent, conditionals/util.py:566-629 behind
 * SeparateIndependentPosterior posteriors.py:863-887), whitened, full q_sqrt [P,m,m]: P covariance pairs into one batched
 * trapezoid, ONE batched factorisation / row-statistics / projection launch sequence instead of the reference's tf.map_fn.
 * family_host [P], variance_host [P], ls_host [P][d] (ard) or [P]: HOST arrays.  Z: latent p's inducing points at
 * Z + p * strideZ (strideZ = 0: shared).  info: P device ints (one factorisation status per latent). */
size_t gpk_svgp_elbo_sep_workspace_bytes(int m, int rows, int d, int P);
int gpk_svgp_elbo_shard_sep(void* stream, const int* family_host, const double* Z, int m, long ldz, long strideZ,
                            const double* Xb, const double* Yb, int rows, long ldxb, long ldyb, int d, int P,
                            const double* ls_host, int ard, const double* variance_host, double noise_variance,
                            const double* noise_rows, double jitter, double mean_const, const double* q_mu,
                            const double* q_sqrt, double* out, int* info, void* ws, size_t ws_bytes);

/* ---- result mailbox in mapped host memory ------------------------------------------------------------------------
 * Copies n doubles from device memory `src` (and one int from `info`, may be NULL) into `host_dst`, a buffer of PINNED,
 * device-mapped host memory (hipHostMalloc / torch pin_memory) laid out as { double vals[n]; int32 info; int32 seq; },
 * then -- after a system-scope release -- writes `seq` last.  The caller spins on host_dst's `seq` word instead of
 * enqueueing device-to-host copies and synchronising the stream: the scalar of SVGP.elbo (svgp.py:181, read by
 * every optimiser step and by monitoring) lands in host memory a few microseconds after the last kernel of the step.
 * (Two blit copies + hipStreamSynchronize cost 90 - 150 us per step on MI355X, profiles/r03_step_timeline.txt.)
 * n <= 16.  One tiny kernel on `stream`; never synchronises. */
int gpk_publish_host(void* stream, const double* src, int n, const int* info, void* host_dst, int seq);

/* Optional per-launch timing of the GEMM kernel (HIP events on the launch stream) for the roofline
 * leg of bench.py: enable(1) starts recording, collect() synchronises and returns the summed kernel
 * time, launch count and ALGORITHMIC flops (useful multiply-adds only) since enable. */
void gpk_profile_gemm_enable(int on);
int gpk_profile_gemm_collect(double* total_ms, long* launches, double* flops);
/* same, restricted to launches with at least min_flops algorithmic flops; keep != 0 keeps the records */
int gpk_profile_gemm_collect_min(double min_flops, int keep, double* total_ms, long* launches, double* flops);
/* same, restricted to launches of ONE kernel -- kind 1 gemm_nt_small, 2 gemm_nt_fast<0,false>, 3 <0,true>, 4 <1,false>,
 * 5 <1,true>, 6 gemm_nt_kernel, 7 the single-launch SVGP step kernel (mega.hip; its flops = the whole step's) -- so that the HIP-event average can be compared with rocprofv3's per-kernel average;
 * records are kept */
int gpk_profile_gemm_collect_kind(int kind, double min_flops, double* total_ms, long* launches, double* flops);
/* phase spanned by the launches with >= min_flops (first start .. last end) and the algorithmic flops of ALL recorded
 * launches issued in between, on any stream: chip-wide rate of a phase in which several streams share the machine */
int gpk_profile_gemm_window(double min_flops, double* window_ms, double* flops_all, double* flops_matching,
                            long* launches_all);

/* micro-benchmarks used by bench.py / profiles (fp64 MFMA issue rate, HBM write stream) */
int gpk_bench_mfma_f64(void* stream, int blocks, int iters, double* sink);
int gpk_bench_stream_store(void* stream, double* out, long n_doubles);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GPK_H */
