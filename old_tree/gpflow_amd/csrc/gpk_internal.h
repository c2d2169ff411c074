// Internal declarations shared by the libgpk translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gpk.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#include <stdio.h>
#include <stdlib.h>

// Tunables.  The PRODUCT build (libgpk.so) has none at run time: every GPK_TUNE is its compile-time default and the
// library never reads the environment.  Only the A/B build (`make exp` -> libgpk_exp.so, -DGPK_EXPERIMENTAL, used by
// tools/ab*.sh on the GPU box and never loaded by the package unless GPK_LIBRARY points at it) reads GPK_<NAME> once.
#ifdef GPK_EXPERIMENTAL
#define GPK_TUNE(name, def)                                                                      \
  ([]() -> int {                                                                                 \
    static const int v__ = getenv("GPK_" #name) ? atoi(getenv("GPK_" #name)) : (int)(def);       \
    return v__;                                                                                  \
  }())
#define GPK_TRACE(...)                                        \
  do {                                                        \
    if (GPK_TUNE(DEBUG, 0)) fprintf(stderr, "[gpk] " __VA_ARGS__); \
  } while (0)
#else
#define GPK_TUNE(name, def) ((int)(def))
#define GPK_TRACE(...) do { } while (0)
#endif

// kGpkExp: host branches that only exist in the A/B build are constant-folded away in the product library.
#ifdef GPK_EXPERIMENTAL
constexpr bool kGpkExp = true;
#else
constexpr bool kGpkExp = false;
#endif

#define GPK_HIP(call)                                                                                   \
  do {                                                                                                  \
    hipError_t e__ = (call);                                                                            \
    if (e__ != hipSuccess) {                                                                            \
      GPK_TRACE("%s:%d: %s -> %d\n", __FILE__, __LINE__, #call, (int)e__);                              \
      return (int)e__;                                                                                  \
    }                                                                                                   \
  } while (0)
#define GPK_LAUNCH_CHECK()                                                                              \
  do {                                                                                                  \
    hipError_t e__ = hipGetLastError();                                                                 \
    if (e__ != hipSuccess) {                                                                            \
      GPK_TRACE("%s:%d: kernel launch -> %d\n", __FILE__, __LINE__, (int)e__);                          \
      return (int)e__;                                                                                  \
    }                                                                                                   \
  } while (0)

static inline size_t gpk_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int gpk_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- GEMM (gemm.hip):  C = alpha * A * B^T + beta * C ----------------------------------------
struct GemmArgs {
  const double* A; long lda; long strideA;   // [m,k]
  const double* B; long ldb; long strideB;   // [n,k]
  double* C; long ldc; long strideC;         // [m,n]
  int m, n, k;
  double alpha, beta;
  int c_lower;      // skip tiles strictly above the diagonal (row r / col c of C: skip if c0 > r_last)
  int b_tri;        // 0 dense, 1 B[j,kk]==0 for kk<j, 2 B[j,kk]==0 for kk>j (+ b_tri_off on kk)
  int a_tri;        // structure of A, a hint that only shortens the K range of a tile: 1 A[i,kk]==0 for kk<i (upper), 2 for kk>i (lower)
  int b_tri_off;    // the triangular structure is B[j,kk] vs kk - b_tri_off
  int b_tri_rows;   // structure applies to rows j < b_tri_rows of B only (rows beyond are dense)
  int k_off_step;   // batch entry z is the K chunk [z k_off_step, z k_off_step + k) of ONE product: a_tri / b_tri refer to the unsplit column index
  // epilogue 1 ("project"): columns < sq_cols are squared and row-summed into part[(tile_n*2+wn), row];
  // columns >= sq_cols (the q_mu rows of the operand) are stored to C2[row, col - sq_cols]; C unused.
  int epi;
  int sq_cols;
  double* part; long part_ld; long stridePart;   // [2*tiles_n, m]
  double* C2; long ldc2; long strideC2; int c2_cols;
  int batch;
  int stagger_first;  // fast path only: number of CUs the launch stream may use (first workgroup of the 2nd resident set), 0 = 256
  int stagger_ticks;  // fast path only: start delay (100 MHz ticks) of the second resident workgroup set, 0 = none
  int no_small;     // never take the one-shot LDS-DMA latency kernel (150 KB of LDS per workgroup: needs a CU free of GEMM workgroups)
  int small_loop;   // K <= 128 launches with MORE than 512 row slivers may still take the one-shot latency kernel: its workgroups
                    // then walk the row blocks with their B tile staged once (the in-group updates of the extra rows)
  int small_kparts; // one-shot latency kernel only: 2 = stage K in two halves (74 KB of LDS per workgroup instead of 146: it then fits
                    // beside a capped bulk workgroup on the same compute unit), else the whole K at once
  int max_wgs;      // fast path only: cap on the number of (persistent) workgroups per batch entry, 0 = one per tile
  int pair_k_align; // set by the launcher for paired triangular-K launches: time-aligned K traversal (gemm_nt_fast)
  // In-kernel stream hand-offs of the factorisation's latency chain (one-shot latency kernel only; potrf.hip, round 5).  An
  // event record / wait between two kernels of one stream costs 4.6 / 6.3 us on MI355X, back-to-back kernels 0.3 us:
  //   sig_ptr:  workgroup (0,0,0) stores sig_val there on entry -- "everything queued before this kernel on its stream has
  //             completed" (in-order queue: the previous kernel's end-of-kernel release is done), read by
  //             hipStreamWaitValue32 on other streams or by another kernel's wait_ptr (every GEMM kernel honours sig_ptr);
  //   wait_ptr: every workgroup spins (bounded) until (int)(*wait_ptr - wait_val) >= 0, then acquires at agent scope: the
  //             word is written by hipStreamWriteValue32 behind the producing kernel on ITS stream.
  int* sig_ptr; int sig_val;
  const int* wait_ptr; int wait_val;
  int* wait_info;   // device int that receives INT_MAX if the bounded wait expires (the factorisation's status word)
  int tile_queue;   // fast path, epi 0: persistent workgroups that take their tiles from a device counter (launches with more than 512 tiles)
  int* queue; int queue_base;   // set by the launcher only: that counter and its value before this launch
  int tile64;       // epi 0 only: take the generic kernel's 64 x 64 tiles (36 KB of LDS per workgroup: fits beside any other workgroup on a CU)
  int tile_snake;   // set by the launcher only (generic kernel, under-filled triangular-K projections): heavy / light tiles alternate per CU
  int tail_first1;  // set by the launcher only (generic 64 x 64 kernel; launch_fast, "tail split"): 1 + first position, 0 = off
};
int gpk_launch_gemm(hipStream_t s, const GemmArgs& a);
bool gpk_gemm_takes_latency_kernel(const GemmArgs& a);   // the launch would run on the one-shot latency kernel (sig / wait honoured)

// fused in-group solve of `rows` right-hand-side rows against nb <= 4 leaf blocks of the factor (gemm.hip); E / Eo point at
// the group's first column, Lgg at L[c0, c0], X at the group's first block inverse
// (batch > 1: blockIdx.y walks the problems, strides in elements)
int gpk_launch_group_solve(hipStream_t s, const double* E, long lde, double* Eo, long ldeo, int rows, const double* Lgg, long ldl,
                           const double* X, int nb, int batch = 1, long strideE = 0, long strideEo = 0, long strideL = 0,
                           long strideX = 0, int max_wgs = 0, int j0 = 0, int j1 = -1);   // max_wgs > 0: at most that many workgroups, walking the 16-row slivers
// fused panel solve + strip of a single-leaf panel (gemm.hip, round 6): P rows below the leaf [m, 128] (solved in place), X the
// leaf's block inverse, C the next block column of the same rows [m, n2]; cnt: two zeroed device words of this launch
bool gpk_panel_fused_ok(const double* P, long lda, const double* X, int m, int nb, int n2);
int gpk_launch_panel_fused(hipStream_t s, double* P, long lda, const double* X, double* C, int m, int n2, int* cnt, int* sig_ptr,
                           int sig_val, const int* wait_ptr, int wait_val, int* wait_info);
int gpk_gemm_tiles_n(int n);   // number of column tiles the launcher will use for n columns
int gpk_profile_gemm_is_on();  // per-launch event timing active (bench roofline leg)
int gpk_prof_begin(hipStream_t s, double flops, int kind);   // same facility for other kernels; returns a record index or -1
void gpk_prof_end(int idx, hipStream_t s);

// ---- single-launch SVGP step (mega.hip): A/B build only (`make exp`, GPK_MEGA=1).  Round 5 applied the stop rule of the
// round-4 review: 2.7 - 3.0 ms at Cm against 2.0 - 2.1 ms for the multi-launch route, so the kernel, its workspace regions and
// its entry points are compiled into libgpk_exp.so only and the product library carries no trace of them.
#ifdef GPK_EXPERIMENTAL
#ifndef GPK_MEGA_DEFAULT
#define GPK_MEGA_DEFAULT 0
#endif
size_t gpk_mega_flag_ints(int m);
int gpk_mega_supported(int m, int rows, int P, int ncu);
int gpk_launch_svgp_mega(hipStream_t s, int proto, int ncu, double* T, long ld, int m, int rows, double* invd, double* Lfin, const double* LqT,
                         long ldl, double* Cacc, const double* q_mu, int P, const double* Y, long ldy, double* s0, double* fmean,
                         double* ssq, double* partial, int* flags, int* info, double* out, double variance, double noise,
                         double mean_const, int min_wgs);
#endif

// ---- leaf (leaf.hip): NB x NB Cholesky + inverse of the diagonal block --------------------------
// A: pointer to the diagonal block (row-major, lda); nb <= NB valid rows/cols.
// sig_ptr: chain-flag word the leaf stores sig_val into on ENTRY ("everything queued before it on s has completed"), or nullptr
int gpk_launch_leaf(hipStream_t s, double* A, long lda, long strideA, int nb, double* invd,
                    long strideInv, int* info, int col0, int batch, int already_factored, int* sig_ptr = nullptr, int sig_val = 0);

// ---- rbf.hip ---------------------------------------------------------------------------------
// (entry point gpk_kernel_matrix is defined there)

// ---- reduce.hip: small kernels -------------------------------------------------------------------
int gpk_launch_zero_upper(hipStream_t s, double* A, int n, long lda, int batch, long strideA);
int gpk_launch_set_identity(hipStream_t s, double* A, int n, long lda, int batch = 1, long strideA = 0);
int gpk_launch_diag_add_scalar(hipStream_t s, double* A, int n, long lda, double v);   // A[i,i] += v
int gpk_probe_concurrent_kernels(hipStream_t a, hipStream_t b, int* scratch, int* concurrent);   // init-time probe (reduce.hip)
int gpk_launch_noop(hipStream_t s);  // empty kernel (stream hand-off probe)
int gpk_launch_wait_flag(hipStream_t s, const int* ptr, int val, int* info);   // one-wave gate: returns when (int)(*ptr - val) >= 0 (bounded)
int gpk_launch_set_flag(hipStream_t s, int* ptr, int val);                     // one-thread store behind everything queued on s
int gpk_launch_sum_parts(hipStream_t s, const double* part, int nt, int rows, long stridePart, int P,
                         double* ssq);
int gpk_launch_final(hipStream_t s, int nterms, const double* const* part, const int* count,
                     const double* scale, double add, double* out);
int gpk_launch_sumsq_stage1(hipStream_t s, const double* A, int rows, int cols, long lda,
                            int upper_only, double* part, int* count);
struct VarexpExtra {   // optional inputs of the variational-expectation stage (reduce.hip, round 6)
  const double* ssq_part = nullptr; int ssq_nt = 0; long ssq_stride = 0;   // ssq as the projection's slot partials [P][nt][rows]
  const int* wait_ptr = nullptr; int wait_val = 0; int* wait_info = nullptr;   // word of the row statistics' stream (bounded wait)
};
int gpk_launch_varexp_stage1(hipStream_t s, const double* Y, long ldy, const double* fmean, int rows,
                             int P, const double* s0, int s0_per_latent, const double* ssq,
                             const double* knn_host, int knn_per_latent, double noise,
                             double mean_const, double* fvar_out, double* part, int* count,
                             const double* noise_rows = nullptr,   // per-row noise variances [rows] or nullptr (constant `noise`)
                             const VarexpExtra* ex = nullptr);
int gpk_launch_kl_white_stage1(hipStream_t s, const double* q_mu, const double* q_sqrt, int m, int P,
                               int q_diag, double* part, int* count);
int gpk_launch_kl_unwhite_diag_stage1(hipStream_t s, const double* LinvT, long ldl, int m, const double* W, int P, double* part,
                                      int* count);
int gpk_launch_sum_log_diag_sq(hipStream_t s, const double* L, int n, long ldl, int batch, long strideL, double* out);
int gpk_launch_row_stats_sep(hipStream_t s, const double* At, long strideAt, int rows, int m, long ldat, const double* V, int P,
                             double* sumsq, double* mv);
int gpk_launch_transpose_shift(hipStream_t s, const double* in, int rows, int cols, long ldin,
                               double* out, long ldout, double shift);
#define GPK_REDUCE_MAXPART 1024
