// Diagonal-block kernel of the blocked Cholesky: one workgroup factors an NB x NB (NB = 128) block
// A = L L^T held entirely in registers and, in the same sweep, forms X = L^-1 (needed by the
// panel solve, which then becomes a plain MFMA GEMM  A21 * X^T).
//
// Replaces the innermost part of tf.linalg.cholesky (gpr.py:102, conditionals/util.py:67, ...).
//
// Layout: 256 threads as a 16 x 16 grid (ti = tid>>4, tj = tid&15); thread (ti,tj) owns the cyclic
// 8x8 sub-lattice  (i = ti + 16a, j = tj + 16b), lower blocks b <= a only (36 slots).  Column step c:
//   1. owners publish the vector u[0..127] to LDS: u[j<c] = X[c][j], u[c] = pivot, u[i>c] = A[i][c]
//   2. one barrier (u is double-buffered across steps)
//   3. every thread: rinv = 1/sqrt(u[c]); then for its rows i > c:
//        slot(i,j) = (j==c ? 0 : slot(i,j)) - (u[i] rinv) * (u[j] rinv)     j <= i
//      which is the Cholesky rank-1 update for j > c and the row operation of the forward
//      substitution  [L | I] -> [I | L^-1]  for j <= c (the slot of the consumed column c is recycled
//      as X[i][c]).  Row c itself becomes the final row of X.
//   4. owners of column c stream l_ic = u[i] rinv to global memory.
// Blocks with nb < 128 are padded with the identity.
#include "gpk_internal.h"

namespace {

constexpr int NB = GPK_NB;
constexpr int SUB = NB / 16;  // 8

template <bool FACTORED>
__global__ __launch_bounds__(256) void leaf_kernel(double* __restrict__ Abase, long lda,
                                                   long strideA, int nb,
                                                   double* __restrict__ invbase, long strideInv,
                                                   int* __restrict__ info, int col0) {
  __shared__ double u[2][NB];
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  double* __restrict__ A = Abase + (long)blockIdx.x * strideA;
  double* __restrict__ inv = invbase + (long)blockIdx.x * strideInv;

  double s[SUB][SUB];
#pragma unroll
  for (int a = 0; a < SUB; ++a)
#pragma unroll
    for (int b = 0; b < SUB; ++b) {
      if (b <= a) {
        const int i = ti + 16 * a, j = tj + 16 * b;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < nb && j < nb && j <= i) v = A[(long)i * lda + j];
        s[a][b] = v;
      } else {
        s[a][b] = 0.0;
      }
    }

  bool bad = false;
  int bad_col = 0;

#pragma unroll
  for (int ca = 0; ca < SUB; ++ca) {
    for (int ct = 0; ct < 16; ++ct) {
      const int c = ca * 16 + ct;
      if (c >= nb) break;  // identity padding: nothing left to do (uniform)
      double* ub = u[c & 1];
      // ---- 1. publish ---------------------------------------------------------------------
      if (ti == ct) {  // row c lives in sub-row ca of these threads
#pragma unroll
        for (int b = 0; b <= ca; ++b) {
          const int j = tj + 16 * b;
          if (j <= c) ub[j] = s[ca][b];
        }
      }
      if (tj == ct) {  // column c lives in sub-column ca
#pragma unroll
        for (int a = ca; a < SUB; ++a) {
          const int i = ti + 16 * a;
          if (i > c) ub[i] = s[a][ca];
        }
      }
      __syncthreads();
      // ---- 3. update ------------------------------------------------------------------------
      const double piv = ub[c];
      double rinv;
      if (FACTORED) {
        rinv = 1.0 / piv;
      } else {
        if (!(piv > 0.0) && !bad) { bad = true; bad_col = c; }
        rinv = 1.0 / sqrt(piv);
      }
      double uj[SUB];
#pragma unroll
      for (int b = 0; b < SUB; ++b) {
        const int j = tj + 16 * b;
        double v = ub[j];
        if (j == c) v = rinv;
        else if (j < c || !FACTORED) v *= rinv;
        uj[b] = v;
      }
#pragma unroll
      for (int a = ca; a < SUB; ++a) {
        const int i = ti + 16 * a;
        double li = ub[i];
        if (!FACTORED) li *= rinv;
        if (i > c) {
#pragma unroll
          for (int b = 0; b <= a; ++b) {
            const int j = tj + 16 * b;
            if (j <= i && !(FACTORED && j > c)) {
              const double base = (j == c) ? 0.0 : s[a][b];
              s[a][b] = base - li * uj[b];
            }
          }
          // ---- 4. stream column c of L ----------------------------------------------------
          if (!FACTORED && tj == ct) A[(long)i * lda + c] = li;
        } else if (i == c) {
#pragma unroll
          for (int b = 0; b <= a; ++b) {
            const int j = tj + 16 * b;
            if (j <= c) s[a][b] = uj[b];
          }
          if (!FACTORED && tj == ct) A[(long)c * lda + c] = piv * rinv;
        }
      }
    }
  }

  // ---- inverse block out (zeros above the diagonal, identity padding kept) --------------------
#pragma unroll
  for (int a = 0; a < SUB; ++a)
#pragma unroll
    for (int b = 0; b < SUB; ++b) {
      const int i = ti + 16 * a, j = tj + 16 * b;
      double v = 0.0;
      if (b <= a && j <= i) v = s[a][b];
      inv[i * NB + j] = v;
    }
  if (!FACTORED && bad && tid == 0 && info) {
    // first failing pivot wins (blocks of one matrix run in stream order)
    if (info[blockIdx.x] == 0) info[blockIdx.x] = col0 + bad_col + 1;
  }
}

}  // namespace

int gpk_launch_leaf(hipStream_t s, double* A, long lda, long strideA, int nb, double* invd,
                    long strideInv, int* info, int col0, int batch, int already_factored) {
  if (nb <= 0 || nb > NB) return GPK_E_ARG;
  dim3 grid((unsigned)(batch > 0 ? batch : 1));
  if (already_factored)
    hipLaunchKernelGGL((leaf_kernel<true>), grid, dim3(256), 0, s, A, lda, strideA, nb, invd,
                       strideInv, info, col0);
  else
    hipLaunchKernelGGL((leaf_kernel<false>), grid, dim3(256), 0, s, A, lda, strideA, nb, invd,
                       strideInv, info, col0);
  GPK_LAUNCH_CHECK();
  return 0;
}
