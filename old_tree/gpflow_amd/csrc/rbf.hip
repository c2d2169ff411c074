// Covariance-matrix builder for stationary kernels (SquaredExponential + Matern family), gfx950.
//
// Replaces, in ONE pass over the output (the reference materialises >= 6 full N x N2 tensors):
//   Stationary.scale            gpflow/kernels/stationaries.py:77-79      X / lengthscales
//   square_distance             gpflow/utilities/ops.py:105-122           ||x||^2 + ||y||^2 - 2 x.y
//   K_r2 / K_r                  stationaries.py:111-116, 209-210, 254-313  sigma^2 k(r)
//   add_noise_cov / Kuu jitter  utilities/model_utils.py:33-38, covariances/kuus.py:33
//
// HBM-write-bound: 64 x 64 output tile per 256-thread workgroup, the two 64 x D input slabs are
// scaled once, transposed into LDS ([D][64]: each thread then reads its 4 rows / 4 cols with
// ds_read_b128 pairs, conflict free) together with their squared norms; each thread produces a
// 4 x 4 patch and stores 32 contiguous bytes per row (16 threads -> 512 B contiguous per row).
// The expansion formula and the association (-2 x.y) + (|x|^2 + |y|^2) of the reference are kept
// so rounding has the same structure (the diagonal is exp(-0.5 * ~1e-16), not exactly sigma^2).
#include "gpk_internal.h"
#include <algorithm>

namespace {

struct RbfArgs {
  const double* X1; long ldx1; int n1;
  const double* X2; long ldx2; int n2;
  int d;
  double* K; long ldk;
  double variance, diag_add;
  int family, sym, lower_only, ard;
  double ls[GPK_MAX_D];
  const double* G; long ldg;  // COMB instantiations only: the output is G .* k(X1, X2) (1) or G + k(X1, X2) (2)
  int comb_diag;              // COMB: X2 is X1 and diag_add goes onto the diagonal of the combined result
  int lds_mirror;             // symmetric full build: write the mirror tile as full rows through LDS
  int nt_store;               // non-temporal stores for complete tiles
};

constexpr int T = 64;

template <int family>
__device__ __forceinline__ double kern_eval(double r2, double variance) {
  if (family == GPK_KERN_SE) return variance * exp(-0.5 * r2);
  const double r = sqrt(fmax(r2, 1e-36));
  if (family == GPK_KERN_MATERN12) return variance * exp(-r);
  if (family == GPK_KERN_MATERN32) {
    const double sqrt3 = 1.7320508075688772;
    return variance * (1.0 + sqrt3 * r) * exp(-sqrt3 * r);
  }
  const double sqrt5 = 2.23606797749979;
  return variance * (1.0 + sqrt5 * r + 5.0 / 3.0 * (r * r)) * exp(-sqrt5 * r);
}

// -2 dk/dr2 at the SCALED squared distance: the factor every lengthscale / input gradient of a stationary kernel starts
// from (for the SquaredExponential it is k itself).  r = sqrt(max(r2, 1e-36)) has derivative 0 where the clamp is
// active (stationaries.py:103-116 under TF autodiff: tf.maximum passes nothing to the clamped argument).
template <int family>
__device__ __forceinline__ double kern_dr2(double r2, double variance) {
  if (family == GPK_KERN_SE) return variance * exp(-0.5 * r2);
  if (!(r2 > 1e-36)) return 0.0;
  const double r = sqrt(r2);
  if (family == GPK_KERN_MATERN12) return variance * exp(-r) / r;
  if (family == GPK_KERN_MATERN32) {
    const double sqrt3 = 1.7320508075688772;
    return 3.0 * variance * exp(-sqrt3 * r);
  }
  const double sqrt5 = 2.23606797749979;
  return (5.0 / 3.0) * variance * (1.0 + sqrt5 * r) * exp(-sqrt5 * r);
}

// MIRROR (symmetric full build): only tiles on or below the diagonal are computed; each is also written
// transposed to its mirror position (K(X,X) from the expansion formula is bitwise symmetric: products and the
// two-term sums commute), which halves the fp64 exp/FMA work of what is otherwise a store-bound kernel.
// COMB = 0: plain build.  COMB = 1 / 2: out = G .* k / G + k, the other factor / term G read from memory (may alias
// the output: every element is read and written by the same thread) -- kernel products and sums (kernels/base.py:216-
// 329) and the elementwise factor of the kernel backward, without materialising the second matrix.
// COMB = 3: out = G .* (-2 dk/dr2); with comb_diag (X2 is X1) the diagonal is written as exact zeros: r2_ii = 0 for
// every parameter value, so it carries no gradient (autodiff of the expansion formula gets rounding noise there,
// amplified by 1/r for Matern12).
template <int FAMILY, int COMB = 0>
__global__ __launch_bounds__(256) void rbf_kernel(RbfArgs p) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int d = p.d;
  double* x1t = sm;                 // [d][T]
  double* x2t = sm + (size_t)d * T; // [d][T]
  double* nr1 = x2t + (size_t)d * T;  // [T]
  double* nr2 = nr1 + T;              // [T]

  const int r0 = blockIdx.y * T, c0 = blockIdx.x * T;
  const bool mirror = p.sym && !p.lower_only;
  if (p.sym && c0 > r0 + T - 1) return;  // strictly upper tile: skipped (lower_only) or written by its mirror
  const int tid = threadIdx.x;

  // stage + scale (true division, as the reference) -- thread t handles (row t>>2, dims (t&3)::4)
  for (int e = tid; e < T * d; e += 256) {
    const int row = e / d, dd = e - row * d;
    const double l = p.ard ? p.ls[dd] : p.ls[0];
    const int g1 = r0 + row, g2 = c0 + row;
    x1t[dd * T + row] = (g1 < p.n1) ? p.X1[(long)g1 * p.ldx1 + dd] / l : 0.0;
    x2t[dd * T + row] = (g2 < p.n2) ? p.X2[(long)g2 * p.ldx2 + dd] / l : 0.0;
  }
  __syncthreads();
  if (tid < 2 * T) {
    const double* src = (tid < T) ? x1t : x2t;
    const int row = tid & (T - 1);
    double s = 0.0;
    for (int dd = 0; dd < d; ++dd) {
      const double v = src[dd * T + row];
      s += v * v;
    }
    ((tid < T) ? nr1 : nr2)[row] = s;
  }
  __syncthreads();

  const int tx = tid & 15, ty = tid >> 4;
  double dot[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dot[i][j] = 0.0;
  for (int dd = 0; dd < d; ++dd) {
    const d2* a2 = reinterpret_cast<const d2*>(&x1t[dd * T + ty * 4]);
    const d2* b2 = reinterpret_cast<const d2*>(&x2t[dd * T + tx * 4]);
    const d2 a01 = a2[0], a23 = a2[1], b01 = b2[0], b23 = b2[1];
    const double a[4] = {a01.x, a01.y, a23.x, a23.y};
    const double b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dot[i][j] = fma(a[i], b[j], dot[i][j]);
  }
  const bool full = (r0 + T <= p.n1) && (c0 + T <= p.n2) && ((p.ldk & 1) == 0) &&
                    ((reinterpret_cast<uintptr_t>(p.K) & 15) == 0);
  double v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gr = r0 + ty * 4 + i;
    const double ni = nr1[ty * 4 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gc = c0 + tx * 4 + j;
      const double r2 = (-2.0 * dot[i][j]) + (ni + nr2[tx * 4 + j]);
      double k = (COMB == 3) ? kern_dr2<FAMILY>(r2, p.variance) : kern_eval<FAMILY>(r2, p.variance);
      if (p.sym && gr == gc) k += p.diag_add;
      if constexpr (COMB != 0) {
        const double g = (gr < p.n1 && gc < p.n2) ? p.G[(long)gr * p.ldg + gc] : 0.0;
        k = (COMB == 2) ? k + g : k * g;
        if (p.comb_diag && gr == gc) k = (COMB == 3) ? 0.0 : k + p.diag_add;
      }
      v[i][j] = k;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gr = r0 + ty * 4 + i;
    if (full) {
      d2* out = reinterpret_cast<d2*>(p.K + (long)gr * p.ldk + c0 + tx * 4);
      if (p.nt_store) {
        __builtin_nontemporal_store((d2){v[i][0], v[i][1]}, out);
        __builtin_nontemporal_store((d2){v[i][2], v[i][3]}, out + 1);
      } else {
        out[0] = (d2){v[i][0], v[i][1]};
        out[1] = (d2){v[i][2], v[i][3]};
      }
    } else if (gr < p.n1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gc = c0 + tx * 4 + j;
        if (gc < p.n2) p.K[(long)gr * p.ldk + gc] = v[i][j];
      }
    }
  }
  if (mirror && r0 != c0 && full && p.lds_mirror) {
    // The transposed copy, written as FULL ROWS: in the register layout a wave's store instruction of the mirror tile
    // touches 16 rows x 4 pieces of 32 B (the lane index runs along the mirror tile's ROWS) -- a quarter of a 128-B line
    // per piece.  Going through LDS ([64][65] doubles, after the input slabs are dead) lets every thread write the same
    // 4 x 32 contiguous bytes per row as in the direct tile: 512 B contiguous per row per 16 lanes.
    __syncthreads();                     // x1t / x2t / norms are no longer read
    double* tl = sm;                     // [T][T + 1]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) tl[(ty * 4 + i) * (T + 1) + tx * 4 + j] = v[i][j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {        // mirror row c0 + ty*4 + i, mirror columns r0 + tx*4 + j  =  S[tx*4 + j][ty*4 + i]
      const int gr = c0 + ty * 4 + i;
      double m[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = tl[(tx * 4 + j) * (T + 1) + ty * 4 + i];
      d2* out = reinterpret_cast<d2*>(p.K + (long)gr * p.ldk + r0 + tx * 4);
      if (p.nt_store) {
        __builtin_nontemporal_store((d2){m[0], m[1]}, out);
        __builtin_nontemporal_store((d2){m[2], m[3]}, out + 1);
      } else {
        out[0] = (d2){m[0], m[1]};
        out[1] = (d2){m[2], m[3]};
      }
    }
  } else if (mirror && r0 != c0) {  // transposed copy: rows c0 + tx*4 + j, columns r0 + ty*4 + i
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = c0 + tx * 4 + j;
      if (full) {
        d2* out = reinterpret_cast<d2*>(p.K + (long)gr * p.ldk + r0 + ty * 4);
        out[0] = (d2){v[0][j], v[1][j]};
        out[1] = (d2){v[2][j], v[3][j]};
      } else if (gr < p.n1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gc = r0 + ty * 4 + i;
          if (gc < p.n2) p.K[(long)gr * p.ldk + gc] = v[i][j];
        }
      }
    }
  }
}

}  // namespace

namespace {
template <int COMB>
int launch_combine(hipStream_t st, int family, const RbfArgs& a, dim3 grid, size_t lds) {
  static const int max_lds = (int)(((size_t)2 * GPK_MAX_D * T + 2 * T) * sizeof(double));
  static const hipError_t at0 = hipFuncSetAttribute(reinterpret_cast<const void*>(rbf_kernel<GPK_KERN_SE, COMB>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  static const hipError_t at1 = hipFuncSetAttribute(reinterpret_cast<const void*>(rbf_kernel<GPK_KERN_MATERN12, COMB>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  static const hipError_t at2 = hipFuncSetAttribute(reinterpret_cast<const void*>(rbf_kernel<GPK_KERN_MATERN32, COMB>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  static const hipError_t at3 = hipFuncSetAttribute(reinterpret_cast<const void*>(rbf_kernel<GPK_KERN_MATERN52, COMB>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  GPK_HIP(at0); GPK_HIP(at1); GPK_HIP(at2); GPK_HIP(at3);
  switch (family) {
    case GPK_KERN_SE: hipLaunchKernelGGL((rbf_kernel<GPK_KERN_SE, COMB>), grid, dim3(256), lds, st, a); break;
    case GPK_KERN_MATERN12: hipLaunchKernelGGL((rbf_kernel<GPK_KERN_MATERN12, COMB>), grid, dim3(256), lds, st, a); break;
    case GPK_KERN_MATERN32: hipLaunchKernelGGL((rbf_kernel<GPK_KERN_MATERN32, COMB>), grid, dim3(256), lds, st, a); break;
    default: hipLaunchKernelGGL((rbf_kernel<GPK_KERN_MATERN52, COMB>), grid, dim3(256), lds, st, a); break;
  }
  GPK_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int gpk_kernel_matrix(void* stream, int family, const double* X1, int n1, long ldx1,
                                 const double* X2, int n2, long ldx2, int d, const double* ls_host,
                                 int ard, double variance, double diag_add, int lower_only,
                                 double* K, long ldk) {
  if (!X1 || !K || !ls_host || n1 < 0 || d <= 0 || d > GPK_MAX_D) return GPK_E_ARG;
  if (family < GPK_KERN_SE || family > GPK_KERN_MATERN52) return GPK_E_UNSUPPORTED;
  RbfArgs a{};
  a.X1 = X1; a.ldx1 = ldx1; a.n1 = n1;
  a.sym = (X2 == nullptr);
  a.X2 = a.sym ? X1 : X2; a.ldx2 = a.sym ? ldx1 : ldx2; a.n2 = a.sym ? n1 : n2;
  a.d = d; a.K = K; a.ldk = ldk; a.variance = variance; a.diag_add = diag_add;
  a.family = family; a.lower_only = lower_only; a.ard = ard;
  for (int i = 0; i < (ard ? d : 1); ++i) a.ls[i] = ls_host[i];
  if (a.n1 == 0 || a.n2 == 0) return 0;
  // (A/B on the full 16384^2 build: mirror tile through LDS 0.479 ms, register layout 0.422 ms; nt stores 0.455 / 0.429)
  a.lds_mirror = GPK_TUNE(RBF_LDS_MIRROR, 0);
  a.nt_store = GPK_TUNE(RBF_NT_STORE, 0);
  size_t lds = ((size_t)2 * d * T + 2 * T) * sizeof(double);
  if (a.sym && !a.lower_only && a.lds_mirror) lds = std::max(lds, (size_t)T * (T + 1) * sizeof(double));
  dim3 grid((unsigned)gpk_cdiv(a.n2, T), (unsigned)gpk_cdiv(a.n1, T));
  return launch_combine<0>((hipStream_t)stream, family, a, grid, lds);
}

// out = G .* k(X1, X2) (op 1) or G + k(X1, X2) (op 2), k recomputed from the inputs, not read: one read of G and one
// write.  Used for (a) the elementwise factor every kernel-parameter gradient starts from (dF/dtheta = sum_ij Kbar_ij
// dK_ij/dtheta and dK/dtheta = K .* (...) for the stationary families) and (b) Product / Sum kernels
// (gpflow/kernels/base.py:216-220, 283-329): the second factor / term is folded into the first matrix in place.
// X2 == NULL means K(X1, X1); diag_add is then added to the diagonal of the COMBINED result.  No symmetric shortcut.

extern "C" int gpk_kernel_matrix_combine(void* stream, int family, int op, const double* X1, int n1, long ldx1,
                                         const double* X2, int n2, long ldx2, int d, const double* ls_host, int ard,
                                         double variance, double diag_add, const double* G, long ldg, double* out,
                                         long ldo) {
  if (!X1 || !G || !out || !ls_host || n1 < 0 || n2 < 0 || d <= 0 || d > GPK_MAX_D) return GPK_E_ARG;
  if (family < GPK_KERN_SE || family > GPK_KERN_MATERN52) return GPK_E_UNSUPPORTED;
  if (op < 1 || op > 3) return GPK_E_ARG;
  RbfArgs a{};
  a.X1 = X1; a.ldx1 = ldx1; a.n1 = n1;
  a.sym = 0;
  a.comb_diag = (X2 == nullptr);
  a.X2 = X2 ? X2 : X1; a.ldx2 = X2 ? ldx2 : ldx1; a.n2 = X2 ? n2 : n1;
  a.d = d; a.K = out; a.ldk = ldo; a.variance = variance; a.diag_add = a.comb_diag ? diag_add : 0.0;
  a.family = family; a.lower_only = 0; a.ard = ard;
  a.G = G; a.ldg = ldg;
  for (int i = 0; i < (ard ? d : 1); ++i) a.ls[i] = ls_host[i];
  if (a.n1 == 0 || a.n2 == 0) return 0;
  const size_t lds = ((size_t)2 * d * T + 2 * T) * sizeof(double);
  dim3 grid((unsigned)gpk_cdiv(a.n2, T), (unsigned)gpk_cdiv(a.n1, T));
  if (op == 3) return launch_combine<3>((hipStream_t)stream, family, a, grid, lds);
  return op == 1 ? launch_combine<1>((hipStream_t)stream, family, a, grid, lds)
                 : launch_combine<2>((hipStream_t)stream, family, a, grid, lds);
}

extern "C" int gpk_kernel_matrix_hadamard(void* stream, int family, const double* X1, int n1, long ldx1,
                                          const double* X2, int n2, long ldx2, int d, const double* ls_host,
                                          int ard, double variance, const double* G, long ldg, double* out,
                                          long ldo) {
  if (!X2) return GPK_E_ARG;
  return gpk_kernel_matrix_combine(stream, family, 1, X1, n1, ldx1, X2, n2, ldx2, d, ls_host, ard, variance, 0.0, G, ldg,
                                   out, ldo);
}
