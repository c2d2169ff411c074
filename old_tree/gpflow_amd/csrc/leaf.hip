// Diagonal-block kernel of the blocked Cholesky (NB = 128): one workgroup (8 waves) factors
// A = L L^T and forms X = L^-1, which turns every panel solve into a plain MFMA GEMM  A21 * X^T.
// Replaces the innermost part of tf.linalg.cholesky (gpr.py:102, conditionals/util.py:67,
// kullback_leiblers.py:107, posteriors.py:422,703).  This kernel IS the critical path of the
// factorisation (serial in the 128 pivots), so it is organised around latency, not throughput.
//
// The block lives in LDS (128 x 130 doubles): L in the lower triangle, the off-diagonal 16x16 tiles of
// X stored TRANSPOSED in the strict upper triangle, the eight diagonal 16x16 tiles of X dense in a side
// buffer.  Right-looking over 16-column sub-blocks k:
//   diag  wave 0 factors the 16x16 diagonal tile and inverts it ENTIRELY IN REGISTERS with
//         v_mfma_f64_16x16x4_f64: a symmetric tile held in the MFMA C/D layout (lane (c,g), reg e <->
//         S[g+4e][c] = S[c][g+4e]) is, register e = p, already the A- and the B-operand of the 4-column
//         panel p, so a panel step is: gather the 4x4 pivot block (v_readlane), factor/invert it with
//         scalar-valued VALU math (the 4 rsqrt chains are the inherent serial part), then four MFMAs
//         (scale the panel, new rows of X, rank-4 update of S, row operations on X) with no data movement.
//   B     all waves:  L_ik = A_ik X_kk^T  (i > k), one 16x16x16 tile product each.
//   C     waves 1..7: A_ij -= L_ik L_jk^T (k < j <= i); wave 0 takes tile (k+1,k+1) first and goes
//         straight on to the next diagonal tile, so the pivot chain overlaps the trailing update.
// Afterwards the off-diagonal part of X is assembled by recursive doubling
//   X21 = -X22 (L21 X11)  at block sizes 16, 32, 64  (log-depth, all tiles of a level in parallel).
// FACTORED = true skips the Cholesky arithmetic and only inverts an existing factor's diagonal block.
#include "gpk_internal.h"
#include "leaf_device.h"
#include "leaf2_device.h"

namespace {
using namespace gpk_leaf;

// The round 1 - 5 leaf described above: still what inverts the diagonal blocks of an existing factor (FACTORED = true) and, with
// FACTORED = false, the A/B baseline of the experimental build (GPK_LEAF_V1=1).  The product factors with leaf2_kernel below.
template <bool FACTORED>
__global__ __launch_bounds__(NT) void leaf_kernel(double* __restrict__ Abase, long lda, long strideA, int nb,
                                                   double* __restrict__ invbase, long strideInv,
                                                   int* __restrict__ info, int col0,
                                                   long long* __restrict__ dbg, int fake_ticks) {
  extern __shared__ __attribute__((aligned(16))) double S[];
#ifdef GPK_EXPERIMENTAL
  if (fake_ticks > 0) {
    // TIMING EXPERIMENT ONLY (GPK_LEAF_FAKE_US): stands in for a leaf of the given duration -- writes L = I and its
    // inverse and spins; results are meaningless, the launch structure of the factorisation is unchanged.
    const long long t0 = wall_clock64();
    double* A = Abase + (long)blockIdx.x * strideA;
    double* inv = invbase + (long)blockIdx.x * strideInv;
    for (int e = threadIdx.x; e < NB * NB; e += NT) {
      const int i = e / NB, j = e % NB;
      if (i < nb && j <= i && j < nb) A[(long)i * lda + j] = (i == j) ? 1.0 : 0.0;
      inv[e] = (i == j) ? 1.0 : 0.0;
    }
    while (wall_clock64() - t0 < fake_ticks) {}
    return;
  }
#endif
  leaf_body<FACTORED>(S, Abase + (long)blockIdx.x * strideA, lda, nb, invbase + (long)blockIdx.x * strideInv,
                      info ? info + blockIdx.x : nullptr, col0, dbg);
}

// the round-6 leaf (leaf2_device.h): twelve waves
__global__ __launch_bounds__(gpk_leaf2::NT2) void leaf2_kernel(double* __restrict__ Abase, long lda, long strideA, int nb,
                                                                double* __restrict__ invbase, long strideInv,
                                                                int* __restrict__ info, int col0, long long* __restrict__ dbg,
                                                                int* __restrict__ sig_ptr, int sig_val) {
  extern __shared__ __attribute__((aligned(16))) double S[];
  // entry signal of the chain flags (potrf.hip): "everything queued before this leaf on the panel stream has completed"
  if (sig_ptr && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(sig_ptr, sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  gpk_leaf2::leaf2_body<false>(S, Abase + (long)blockIdx.x * strideA, lda, nb, invbase + (long)blockIdx.x * strideInv,
                               info ? info + blockIdx.x : nullptr, col0, dbg);
}

}  // namespace

#ifdef GPK_EXPERIMENTAL
namespace {
constexpr int DBG_CAP = 4096;
long long* g_dbg = nullptr;
int g_dbg_n = 0;
int g_dbg_col[DBG_CAP];
}  // namespace
extern "C" __attribute__((visibility("default"))) int gpk_exp_leaf_dbg_dump(int first) {
  if (!g_dbg || g_dbg_n <= first) return 0;
  static long long host[8 * DBG_CAP];
  if (hipMemcpy(host, g_dbg, sizeof(long long) * 8 * g_dbg_n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  printf("# leaf phases in us (100 MHz wall clock): col0 load factor invert store total | begin since the first leaf, gap since the previous leaf's end\n");
  for (int i = first; i < g_dbg_n; ++i)
    printf("leaf %5d  %6.1f %6.1f %6.1f %6.1f  %6.1f | %9.1f %7.1f\n", g_dbg_col[i], host[8 * i] / 100.0, host[8 * i + 1] / 100.0,
           host[8 * i + 2] / 100.0, host[8 * i + 3] / 100.0, host[8 * i + 4] / 100.0, (host[8 * i + 5] - host[8 * first + 5]) / 100.0,
           i > first ? (host[8 * i + 5] - host[8 * (i - 1) + 5] - host[8 * (i - 1) + 4]) / 100.0 : 0.0);
  fflush(stdout);
  const int n = g_dbg_n;
  g_dbg_n = 0;
  return n;
}
#endif

int gpk_launch_leaf(hipStream_t s, double* A, long lda, long strideA, int nb, double* invd,
                    long strideInv, int* info, int col0, int batch, int already_factored, int* sig_ptr, int sig_val) {
  if (nb <= 0 || nb > NB) return GPK_E_ARG;
  // (function-local statics: initialised once, thread-safe)
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(leaf_kernel<true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS);
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(leaf_kernel<false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS);
  static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(leaf2_kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)gpk_leaf2::LEAF2_LDS);
  GPK_HIP(attr1);
  GPK_HIP(attr0);
  GPK_HIP(attr2);
  dim3 grid((unsigned)(batch > 0 ? batch : 1));
  const int fake = kGpkExp ? GPK_TUNE(LEAF_FAKE_US, 0) * 100 : 0;
  const int v1 = kGpkExp ? (GPK_TUNE(LEAF_V1, 0) || fake > 0) : 0;
  if (sig_ptr && (already_factored || v1)) {   // (only the round-6 kernel carries the entry signal)
    const int rcs = gpk_launch_set_flag(s, sig_ptr, sig_val);
    if (rcs) return rcs;
    sig_ptr = nullptr;
  }
#ifdef GPK_EXPERIMENTAL
  // phase timers of every leaf launch (GPK_LEAF_DBG=1; printed by gpk_exp_leaf_dbg_dump): load / factor / invert / store
  if (!already_factored && GPK_TUNE(LEAF_DBG, 0) && g_dbg_n < DBG_CAP) {
    if (!g_dbg) GPK_HIP(hipMalloc(&g_dbg, sizeof(long long) * 8 * DBG_CAP));
    g_dbg_col[g_dbg_n] = col0;
    if (v1)
      hipLaunchKernelGGL((leaf_kernel<false>), grid, dim3(NT), LEAF_LDS, s, A, lda, strideA, nb, invd, strideInv, info, col0,
                         g_dbg + 8 * (g_dbg_n++), fake);
    else
      hipLaunchKernelGGL(leaf2_kernel, grid, dim3(gpk_leaf2::NT2), gpk_leaf2::LEAF2_LDS, s, A, lda, strideA, nb, invd, strideInv, info,
                         col0, g_dbg + 8 * (g_dbg_n++), sig_ptr, sig_val);
    GPK_LAUNCH_CHECK();
    return 0;
  }
#endif
  if (already_factored)
    hipLaunchKernelGGL((leaf_kernel<true>), grid, dim3(NT), LEAF_LDS, s, A, lda, strideA, nb, invd,
                       strideInv, info, col0, nullptr, 0);
  else if (v1)
    hipLaunchKernelGGL((leaf_kernel<false>), grid, dim3(NT), LEAF_LDS, s, A, lda, strideA, nb, invd,
                       strideInv, info, col0, nullptr, fake);
  else
    hipLaunchKernelGGL(leaf2_kernel, grid, dim3(gpk_leaf2::NT2), gpk_leaf2::LEAF2_LDS, s, A, lda, strideA, nb, invd, strideInv, info,
                       col0, nullptr, sig_ptr, sig_val);
  GPK_LAUNCH_CHECK();
  return 0;
}
