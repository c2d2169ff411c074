// Right-side triangular solve of a row panel against one diagonal GROUP of the factor, in ONE launch:
//     Eout[:, 0:gw] = Ein[:, 0:gw] * L_gg^-T          (gw = 128 * nb <= 512 columns, L_gg lower, nb leaf blocks)
// This is the tf.linalg.triangular_solve(Lm, Kmn) of conditionals/util.py:125 restricted to a column group of
// A^T = Kfu Lm^-T.  Blocked substitution with the leaf kernel's inverses X_jj of the 128x128 diagonal blocks:
//     for j = 0 .. nb-1:   E_j <- E_j X_jj^T ;   E_j' -= E_j L_j'j^T  (j' > j)
// The rows of E are independent, so a workgroup owns 16 rows for the whole substitution: its 16 x gw slab lives
// in LDS (A operands, accumulator tiles), the B operands (X_jj, L_j'j: L2-resident, shared by every workgroup) go
// global -> VGPR directly in MFMA fragment order.  Seven dependent phases inside one kernel replace seven dependent
// kernel launches (each of which filled a quarter of the chip for 20-60 us).  In place (Eout == Ein) is fine.
#include "gpk_internal.h"

namespace {
constexpr int NB = GPK_NB;
constexpr int TG_ROWS = 16, TG_THREADS = 512, TG_WAVES = 8;

__device__ __forceinline__ d4 mfma4(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// one 16x16 output tile over K = 128:  acc (+)= sign * Es[:, acol0 + k] * B[n, k],  B row-major [16 rows n][ldb]
template <bool NEG>
__device__ __forceinline__ d4 tile_k128(const double* __restrict__ Es, int ldl, int acol0, const double* __restrict__ Bt,
                                        long ldb, int lane, d4 acc) {
  const int r = lane & 15, kq = lane >> 4;
  double b[32];
  const double* bp = Bt + (long)r * ldb + kq;
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) b[kk] = bp[4 * kk];
  const double* ap = Es + r * ldl + acol0 + kq;
  d4 acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 32; kk += 2) {
    const double a0 = ap[4 * kk], a1 = ap[4 * kk + 4];
    acc = mfma4(NEG ? -a0 : a0, b[kk], acc);
    acc2 = mfma4(NEG ? -a1 : a1, b[kk + 1], acc2);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] += acc2[e];
  return acc;
}

__global__ __launch_bounds__(TG_THREADS) void trsm_group_kernel(const double* __restrict__ Ein, long ldein,
                                                                 double* __restrict__ Eout, long ldeout, int rows,
                                                                 const double* __restrict__ Lgg, long lda,
                                                                 const double* __restrict__ invg, int nb) {
  extern __shared__ __attribute__((aligned(16))) double Es[];  // [16][ldl]
  const int gw = nb * NB, ldl = gw + 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * TG_ROWS;
  // ---- load the 16 x gw slab (rows clamped; out-of-range rows are never stored) -------------------------------
  {
    const int cpr = gw / 2;  // 16-byte chunks per row
    const bool vec = ((ldein & 1) == 0) && ((reinterpret_cast<uintptr_t>(Ein) & 15) == 0);
    for (int c = tid; c < TG_ROWS * cpr; c += TG_THREADS) {
      const int r = c / cpr, q = c - r * cpr;
      int gr = row0 + r;
      gr = gr < rows ? gr : rows - 1;
      const double* src = Ein + (long)gr * ldein + 2 * q;
      d2 v;
      if (vec) v = *reinterpret_cast<const d2*>(src);
      else { v.x = src[0]; v.y = src[1]; }
      *reinterpret_cast<d2*>(&Es[r * ldl + 2 * q]) = v;
    }
  }
  __syncthreads();
  const int c = lane & 15, g = lane >> 4;
  for (int jb = 0; jb < nb; ++jb) {
    // ---- solve: E_j <- E_j X_jj^T ; output tile `wave` of the block (16 columns), K = 128 ---------------------
    d4 r = tile_k128<false>(Es, ldl, jb * NB, invg + (long)jb * NB * NB + (long)(16 * wave) * NB, NB, lane,
                            (d4){0.0, 0.0, 0.0, 0.0});
    __syncthreads();  // every wave has read block j as its A operand
#pragma unroll
    for (int e = 0; e < 4; ++e) Es[(g + 4 * e) * ldl + jb * NB + 16 * wave + c] = r[e];
    __syncthreads();
    // ---- update: E_j' -= E_j L_j'j^T for j' > j ; (nb-1-jb) * 8 tiles over the 8 waves -------------------------
    const int ntile = (nb - 1 - jb) * 8;
    for (int t = wave; t < ntile; t += TG_WAVES) {
      const int jp = jb + 1 + (t >> 3), v = t & 7;
      const int col0 = jp * NB + 16 * v;
      d4 acc;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = Es[(g + 4 * e) * ldl + col0 + c];
      acc = tile_k128<true>(Es, ldl, jb * NB, Lgg + (long)(jp * NB + 16 * v) * lda + jb * NB, lda, lane, acc);
#pragma unroll
      for (int e = 0; e < 4; ++e) Es[(g + 4 * e) * ldl + col0 + c] = acc[e];
    }
    __syncthreads();
  }
  // ---- store ---------------------------------------------------------------------------------------------------
  {
    const int cpr = gw / 2;
    const bool vec = ((ldeout & 1) == 0) && ((reinterpret_cast<uintptr_t>(Eout) & 15) == 0);
    for (int cidx = tid; cidx < TG_ROWS * cpr; cidx += TG_THREADS) {
      const int r = cidx / cpr, q = cidx - r * cpr;
      const int gr = row0 + r;
      if (gr >= rows) continue;
      const d2 v = *reinterpret_cast<const d2*>(&Es[r * ldl + 2 * q]);
      double* dst = Eout + (long)gr * ldeout + 2 * q;
      if (vec) *reinterpret_cast<d2*>(dst) = v;
      else { dst[0] = v.x; dst[1] = v.y; }
    }
  }
}
}  // namespace

// Ein/Eout point at column g0 of the row panel; Lgg at L[g0][g0]; invg at the inverse of leaf block g0/128.
int gpk_launch_trsm_group(hipStream_t s, const double* Ein, long ldein, double* Eout, long ldeout, int rows,
                          const double* Lgg, long lda, const double* invg, int nb) {
  if (rows <= 0 || nb <= 0) return 0;
  if (nb > 4) return GPK_E_ARG;
  const size_t lds = (size_t)TG_ROWS * (nb * NB + 2) * sizeof(double);
  dim3 grid((unsigned)gpk_cdiv(rows, TG_ROWS));
  hipLaunchKernelGGL(trsm_group_kernel, grid, dim3(TG_THREADS), lds, s, Ein, ldein, Eout, ldeout, rows, Lgg, lda, invg,
                     nb);
  GPK_LAUNCH_CHECK();
  return 0;
}
