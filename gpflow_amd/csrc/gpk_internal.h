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

// Experimental device code (ticketed GEMM tiles under a software CU reservation, the tile-dataflow bulk kernel, the CU
// census) exists only in the A/B build: each of these was built, is parity-tested through the A/B library and measured
// SLOWER than the default scheme on MI355X (DESIGN.md section 6 has the numbers and the reasons).  The product library
// contains none of those kernels; the host code that would drive them is constant-folded away (kGpkExp).
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
  int max_wgs;      // fast path only: cap on the number of (persistent) workgroups per batch entry, 0 = one per tile
  // Ticketed launch (fast path, batch 1): workgroups draw tiles from per-XCD counters instead of owning a fixed tile
  // list, and a workgroup that finds itself on a compute unit listed in `resv` takes none and exits -- the SOFTWARE
  // CU reservation that keeps the latency chain of a factorisation (leaf / panel solve / strip) dispatchable while a
  // bulk GEMM owns the rest of the chip.  ctr: 16 zeroed device words owned by the launch stream (self-resetting).
  unsigned* ctr;
  const unsigned char* resv;   // [GPK_CU_KEYS] 1 = reserved, or NULL
  int c_l1_bypass;  // fast tile: preload C with L1-bypassing (device-scope, sc1) loads -- C may have been rewritten by another CU of this XCD
                    // since this CU last saw it (dataflow kernel: read-modify-write tiles change hands between tasks)
};
#define GPK_CU_KEYS 4096   // key = XCC_ID << 8 | HW_ID[15:8] (SE_ID, SH_ID, CU_ID)
// census of the physical compute units (gemm.hip): fills keys[0..n) with the key of the CU each of n workgroups ran on
int gpk_cu_census(hipStream_t s, unsigned* keys_dev, int n);
int gpk_launch_gemm(hipStream_t s, const GemmArgs& a);

// ---- tile-dataflow bulk kernel (gemm.hip; task lists: flow_tasks.h) ---------------------------------------------------
#include "flow_tasks.h"
struct FlowArgs {
  double* E; long lde;                 // minibatch rows, consumed: [rows, n]
  double* Eo; long ldeo;               // solved rows A^T = E L^-T: [rows, n]
  const double* L; long ldl;           // the factor being built (columns of group g are final once flag g is raised)
  const double* invd;                  // leaf block inverses [n / 128][128][128]
  const double* gws;                   // explicit group inverses: group i lives at gws + i * 2 * 512 * 512 + 512 * 512, leading dimension = its width
  const double* LqT; long ldq; long strideQ;    // [P][n, ldq] tril(q_sqrt)^T, or NULL (no projection tasks in the list)
  double* Cacc; long ldc; long strideC;          // [P][rows, ldc] running A^T Lq
  double* part; long part_ld; long stridePart;  // [P][2 * n / 128, rows] partial row sums of squares
  int rows, n, P, ng;
  int g0[GPK_FLOW_MAX_GROUPS], g1[GPK_FLOW_MAX_GROUPS], ginv[GPK_FLOW_MAX_GROUPS];
  const FlowTask* tasks; int off[9];   // per-XCD ticket lists: tasks[off[x] .. off[x + 1])
  unsigned* ctr;                       // [16] tickets per XCD, workgroups that left / took part (zeroed before the launch)
  unsigned* prog;                      // [rows / 128] finished tasks per row block (zeroed before the launch)
  const unsigned* flags;               // [ng] chain flags: == epoch once the group's columns (and inverse) exist
  unsigned epoch;
  const unsigned char* resv;           // software CU reservation table or NULL
  int* info;                           // factorisation status word: set to 0x40000000 if a bounded wait expires
  int coh;                             // coherence protocol between tasks (see flow_kernel)
};
int gpk_launch_flow(hipStream_t s, const FlowArgs& f);
int gpk_launch_set_flag(hipStream_t s, unsigned* flag, unsigned value);
int gpk_gemm_tiles_n(int n);   // number of column tiles the launcher will use for n columns
int gpk_profile_gemm_is_on();  // per-launch event timing active (bench roofline leg)

// ---- leaf (leaf.hip): NB x NB Cholesky + inverse of the diagonal block --------------------------
// A: pointer to the diagonal block (row-major, lda); nb <= NB valid rows/cols.
int gpk_launch_leaf(hipStream_t s, double* A, long lda, long strideA, int nb, double* invd,
                    long strideInv, int* info, int col0, int batch, int already_factored);

// ---- chain panel (leaf.hip, A/B build): ONE launch per 128-column panel of an SVGP-sized factorisation ------------
// workgroup 0 = the leaf of panel p; 8 helpers per row block solve the rows of blocks p+1 and p+2 against the new
// inverse and apply panel p to the three tiles the NEXT TWO leaves depend on -- (p+1,p+1), (p+2,p+1), (p+2,p+2) -- with
// in-kernel flags instead of launch boundaries; everything else of panel p (the other rows' solve, strips, rest-update)
// is bulk work on other streams, released by hipStreamWaitValue32 on flagK.
struct LeafKArgs {
  double* A; long lda;       // the whole trapezoid (row-major)
  int nblk;                  // number of 128-column blocks of the square part (n = 128 nblk)
  int p;                     // panel
  double* invd;              // [nblk][128][128] block inverses (block p written here)
  int* info;
  unsigned* flagL;           // device word: leaf of this launch done (agent scope)
  unsigned long long* cnt;   // [2] device counters: helpers past the solve / helpers finished (zeroed by workgroup 0)
  unsigned* flagK;           // signal memory: the critical tiles of this launch are done (system scope)
  const unsigned* flagB;     // signal memory (hipStreamWriteValue32): strips of the previous panel done
  unsigned epoch;            // value raised on flagL / flagK
  unsigned need_b;           // flagB must have reached this value (0: nothing to wait for)
  int nh;                    // helpers: 8 (block p+1) + 8 (block p+2) + 8 (tile (p+2,p+2)), fewer at the end
  int nsolve;                // of which solve a sliver (the first 8 or 16)
};
int gpk_launch_leafk(hipStream_t s, const LeafKArgs& a);

// ---- rbf.hip ---------------------------------------------------------------------------------
// (entry point gpk_kernel_matrix is defined there)

// ---- reduce.hip: small kernels -------------------------------------------------------------------
int gpk_launch_zero_upper(hipStream_t s, double* A, int n, long lda, int batch, long strideA);
int gpk_launch_set_identity(hipStream_t s, double* A, int n, long lda);
int gpk_launch_sum_parts(hipStream_t s, const double* part, int nt, int rows, long stridePart, int P,
                         double* ssq);
int gpk_launch_final(hipStream_t s, int nterms, const double* const* part, const int* count,
                     const double* scale, double add, double* out);
int gpk_launch_sumsq_stage1(hipStream_t s, const double* A, int rows, int cols, long lda,
                            int upper_only, double* part, int* count);
int gpk_launch_varexp_stage1(hipStream_t s, const double* Y, long ldy, const double* fmean, int rows,
                             int P, const double* s0, int s0_per_latent, const double* ssq,
                             const double* knn_host, int knn_per_latent, double noise,
                             double mean_const, double* fvar_out, double* part, int* count);
int gpk_launch_kl_white_stage1(hipStream_t s, const double* q_mu, const double* q_sqrt, int m, int P,
                               int q_diag, double* part, int* count);
int gpk_launch_transpose_shift(hipStream_t s, const double* in, int rows, int cols, long ldin,
                               double* out, long ldout, double shift);
#define GPK_REDUCE_MAXPART 1024
