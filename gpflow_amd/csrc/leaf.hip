// Diagonal-block kernel of the blocked Cholesky (NB = 128): one workgroup (4 waves) factors
// A = L L^T and, in the same sweep, forms X = L^-1, which turns the panel solve into a plain MFMA
// GEMM  A21 * X^T.  Replaces the innermost part of tf.linalg.cholesky (gpr.py:102,
// conditionals/util.py:67, kullback_leiblers.py:107, posteriors.py:422,703).
//
// The whole block lives in LDS (128 x 130 doubles = 130 KB): L in the lower triangle, the running
// inverse stored TRANSPOSED in the strict upper triangle (exactly the [n][k] layout the NT MFMA
// fragments want) and its diagonal in a side array.  Right-looking over 16-column sub-blocks k:
//   A. wave 0: 16x16 Cholesky + inverse of the diagonal sub-block in registers (lane = row,
//      cross-lane traffic by v_readlane; rsqrt = v_rsq_f64 + 2 Newton steps) -- the serial chain;
//   B. all waves (v_mfma_f64_16x16x4): X_kj = X_kk W_kj (j<k) and L_ik = A_ik X_kk^T (i>k);
//   C. all waves: for i>k:  A_ij -= L_ik L_jk^T (k<j<=i),  W_ik = -L_ik X_kk,  W_ij -= L_ik X_kj (j<k)
// i.e. the row operations of [L | I] -> [I | L^-1] ride along with the trailing update.
// FACTORED = true skips the Cholesky arithmetic and only inverts an existing factor's diagonal block.
#include "gpk_internal.h"

namespace {

constexpr int NB = GPK_NB;
constexpr int LD = NB + 2;
constexpr int SB = 16;        // sub-block
constexpr int NT = 512;       // threads per workgroup (8 waves: 2 per SIMD hide LDS/MFMA latency)
constexpr size_t LEAF_LDS = ((size_t)NB * LD + NB + 2) * sizeof(double);

__device__ __forceinline__ double readlane_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

__device__ __forceinline__ double rsqrt_nr(double p) {
  double y = __builtin_amdgcn_rsq(p);
  const double h = 0.5 * p;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  return y;
}


// broadcast lane J (of each row of 16 lanes) to the whole row: DPP row_share, no SGPR round trip
template <int J, bool NOP>
__device__ __forceinline__ double row_share_d(double v) {
  // one 64-bit DPP move (gfx90a+ "DP ALU DPP", row_newbcast only).  NOP: cover the
  // VALU-write -> DPP-read hazard on the first move of a batch (hipcc does not pad inline asm).
  double out;
  if constexpr (NOP)
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "=v"(out) : "v"(v), "n"(J));
  else
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "=v"(out) : "v"(v), "n"(J));
  return out;
}

// batch the broadcasts, then the FMAs: a dependent DPP->FMA pair would stall the in-order wave
template <int C, int J>
struct LBcast {  // lj[j] = l of lane j, j = J..15
  static __device__ __forceinline__ void run(double (&lj)[16], double l) {
    if constexpr (J < 16) {
      lj[J] = row_share_d<J, (J == C + 1)>(l);
      LBcast<C, J + 1>::run(lj, l);
    }
  }
};
template <int C, int J>
struct XBcast {  // xc[j] = x[j] of lane C, j = J..C-1
  static __device__ __forceinline__ void run(double (&xc)[16], const double (&x)[16]) {
    if constexpr (J < C) {
      xc[J] = row_share_d<C, (J == 0)>(x[J]);
      XBcast<C, J + 1>::run(xc, x);
    }
  }
};

// one column step of the 16x16 diagonal sub-block (all lanes; lane&15 = row, rows mirrored x4)
template <bool FACTORED, int C>
struct DiagStep {
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double& myrinv, int row,
                                             int& bad_col, int kb) {
    if constexpr (C < 16) {
      const double p = row_share_d<C, true>(a[C]);
      double xc[16];
      XBcast<C, 0>::run(xc, x);  // independent of the rsqrt chain: issue first
      double rinv, l;
      if constexpr (FACTORED) {
        rinv = 1.0 / p;
        l = a[C];
      } else {
        if (!(p > 0.0) && bad_col < 0) bad_col = kb + C;
        rinv = rsqrt_nr(p);
        l = a[C] * rinv;
        a[C] = l;
      }
      myrinv = (row == C) ? rinv : myrinv;
      const double le = (row > C) ? l : 0.0;
      if constexpr (!FACTORED) {
        double lj[16];
        LBcast<C, C + 1>::run(lj, l);
#pragma unroll
        for (int j = C + 1; j < 16; ++j) a[j] = fma(-le, lj[j], a[j]);
      }
      const double ler = le * rinv;
#pragma unroll
      for (int j = 0; j < C; ++j) x[j] = fma(-ler, xc[j], x[j]);
      x[C] = -ler;
      DiagStep<FACTORED, C + 1>::run(a, x, myrinv, row, bad_col, kb);
    }
  }
};

__device__ __forceinline__ d4 mfma4(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- 16x16x16 block-product tasks on the LDS-resident matrix ------------------------------------------
// acc[e] (m = g + 4e, n = r)  =  (init ? S[c[e]] : 0)  +  sum_q (neg ? -1 : 1) * S[a[kk]] * S[b[kk]]
// All operands are plain LDS offsets (the triangular X_kk accessor resolves to an offset too, with a
// dedicated zero slot), so tasks are uniform and software-pipelined: the operands of task t+1 are in
// flight while the 4 MFMAs of task t issue.
constexpr int ZERO_SLOT = NB * LD + NB;  // S[ZERO_SLOT] == 0.0

struct Task {
  int a[4], b[4], c[4];
  bool neg, init;
};

// offset of X_kk[row][col] (lower; stored transposed in the strict upper part + diagonal in xd)
__device__ __forceinline__ int xkk_off(int k, int row, int col) {
  const int b = k * SB;
  if (col < row) return (b + col) * LD + b + row;
  if (col == row) return NB * LD + b + row;
  return ZERO_SLOT;
}

struct TaskRegs {
  double a[4], b[4], c[4];
};

__device__ __forceinline__ void task_load(const double* S, const Task& T, TaskRegs& R) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    R.a[kk] = S[T.a[kk]];
    R.b[kk] = S[T.b[kk]];
    R.c[kk] = T.init ? S[T.c[kk]] : 0.0;
  }
}

__device__ __forceinline__ void task_exec(double* S, const Task& T, const TaskRegs& R) {
  d4 acc = {R.c[0], R.c[1], R.c[2], R.c[3]};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = mfma4(T.neg ? -R.a[kk] : R.a[kk], R.b[kk], acc);
#pragma unroll
  for (int e = 0; e < 4; ++e) S[T.c[e]] = acc[e];
}

template <class F>
__device__ __forceinline__ void run_tasks(double* S, int wave, int ntask, F make) {
  constexpr int NWV = NT / 64;
  Task T0, T1;
  TaskRegs R0, R1;
  int t = wave;
  if (t >= ntask) return;
  make(t, T0);
  task_load(S, T0, R0);
  for (;;) {
    const int t1 = t + NWV;
    const bool has1 = t1 < ntask;
    if (has1) { make(t1, T1); task_load(S, T1, R1); }
    task_exec(S, T0, R0);
    if (!has1) break;
    const int t2 = t1 + NWV;
    const bool has2 = t2 < ntask;
    if (has2) { make(t2, T0); task_load(S, T0, R0); }
    task_exec(S, T1, R1);
    if (!has2) break;
    t = t2;
  }
}

template <bool FACTORED>
__global__ __launch_bounds__(NT) void leaf_kernel(double* __restrict__ Abase, long lda,
                                                   long strideA, int nb,
                                                   double* __restrict__ invbase, long strideInv,
                                                   int* __restrict__ info, int col0,
                                                   long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) double S[];  // [NB][LD] then xd[NB]
  double* xd = S + NB * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  double* __restrict__ A = Abase + (long)blockIdx.x * strideA;
  double* __restrict__ inv = invbase + (long)blockIdx.x * strideInv;
  const int nsb = (nb + SB - 1) / SB;
  const int npad = nsb * SB;

  const long long t_load0 = dbg ? wall_clock64() : 0;
  // ---- load: lower triangle of A (identity beyond nb), zero strict upper -----------------------
  {
    // thread t owns column pair (2*(t&63), +1) of rows (t>>6) + NW*it: 64 lanes read one 1 KiB row
    const int jp = 2 * (tid & 63), r0 = tid >> 6;
    const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    constexpr int NW = NT / 64, NIT = NB / NW;
#pragma unroll 8
    for (int it = 0; it < NIT; ++it) {
      const int i = r0 + NW * it;
      d2 v = {0.0, 0.0};
      if (i < nb && jp <= i) {
        const double* src = A + (long)i * lda + jp;
        if (vec && jp + 1 < nb) v = *reinterpret_cast<const d2*>(src);
        else { v.x = src[0]; if (jp + 1 < nb) v.y = src[1]; }
      }
      if (jp > i) v.x = 0.0; else if (i >= nb) v.x = (jp == i) ? 1.0 : 0.0;
      if (jp + 1 > i) v.y = 0.0; else if (i >= nb) v.y = (jp + 1 == i) ? 1.0 : 0.0;
      *reinterpret_cast<d2*>(&S[i * LD + jp]) = v;
    }
  }
  if (tid == 0) { S[NB * LD + NB] = 0.0; S[NB * LD + NB + 1] = 0.0; }
  __syncthreads();
  long long tA = 0, tB = 0, tC = 0, t_prev = wall_clock64();
  const long long t_start = t_prev;
  const long long c_start = clock64();
  if (dbg && tid == 0) dbg[0] = t_prev - t_load0;

  int bad_col = -1;

  for (int k = 0; k < nsb; ++k) {
    const int kb = k * SB;
    // ================= A: diagonal sub-block, wave 0 ================================================
    if (wave == 0) {
      double a[SB], x[SB];
      const int row = lane & 15;  // lanes 16..63 mirror lanes 0..15 (each DPP row is a full copy)
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        a[j] = (j <= row) ? S[(kb + row) * LD + kb + j] : 0.0;
        x[j] = 0.0;
      }
      double myrinv = 0.0;
      DiagStep<FACTORED, 0>::run(a, x, myrinv, row, bad_col, kb);
      if (lane < SB) {
#pragma unroll
        for (int j = 0; j < SB; ++j) {
          if (!FACTORED && j <= row) S[(kb + row) * LD + kb + j] = a[j];
          if (j < row) S[(kb + j) * LD + kb + row] = x[j] * myrinv;  // X[row][j], transposed
        }
        xd[kb + row] = myrinv;
      }
    }
    __syncthreads();
    if (dbg) { const long long t = wall_clock64(); tA += t - t_prev; t_prev = t; }
    // ================= B: X_kj = X_kk W_kj (j<k);  L_ik = A_ik X_kk^T (i>k) ===========================
    {
      const int ntask = FACTORED ? k : (nsb - 1);  // j in [0,k) then i in (k, nsb)
      run_tasks(S, wave, ntask, [&](int t, Task& T) {
        T.neg = false; T.init = false;
        if (t < k) {
          const int jb = t * SB;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int q = 4 * kk + g;
            T.a[kk] = xkk_off(k, r, q);
            T.b[kk] = (jb + r) * LD + kb + q;
            T.c[kk] = (jb + r) * LD + kb + g + 4 * kk;  // X_kj[m][n] -> S[jb + n][kb + m]
          }
        } else {
          const int ib = (t + 1) * SB;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int q = 4 * kk + g;
            T.a[kk] = (ib + r) * LD + kb + q;
            T.b[kk] = xkk_off(k, r, q);
            T.c[kk] = (ib + g + 4 * kk) * LD + kb + r;
          }
        }
      });
    }
    __syncthreads();
    if (dbg) { const long long t = wall_clock64(); tB += t - t_prev; t_prev = t; }
    // ================= C: trailing update + inverse row operations =======================================
    {
      // tasks: rows i in (k, nsb), cols j in [0, i] (FACTORED: j in [0, k])
      //   j<k: W_ij -= L_ik X_kj    j==k: W_ik = -L_ik X_kk    j>k: A_ij -= L_ik L_jk^T
      const int rows_below = nsb - 1 - k;
      int ntask;
      if (FACTORED) ntask = rows_below * (k + 1);
      else ntask = (nsb * (nsb + 1)) / 2 - ((k + 1) * (k + 2)) / 2;
      run_tasks(S, wave, ntask, [&](int t, Task& T) {
        int i, j;
        if (FACTORED) {
          i = k + 1 + t / (k + 1);
          j = t - (i - k - 1) * (k + 1);
        } else {
          i = k + 1;
          int rem = t;
          while (rem > i) { rem -= i + 1; ++i; }
          j = rem;
        }
        const int ib = i * SB, jb = j * SB;
        T.neg = true;
        T.init = (j != k);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int q = 4 * kk + g;
          T.a[kk] = (ib + r) * LD + kb + q;
          if (j == k) {
            T.b[kk] = xkk_off(k, q, r);                  // Bop[n][q] = X_kk[q][n]
            T.c[kk] = (kb + r) * LD + ib + g + 4 * kk;   // W_ik^T
          } else if (j < k) {
            T.b[kk] = (jb + r) * LD + kb + q;            // X_kj^T
            T.c[kk] = (jb + r) * LD + ib + g + 4 * kk;   // W_ij^T
          } else {
            T.b[kk] = (jb + r) * LD + kb + q;            // L_jk
            T.c[kk] = (ib + g + 4 * kk) * LD + jb + r;   // A_ij
          }
        }
      });
    }
    __syncthreads();
    if (dbg) { const long long t = wall_clock64(); tC += t - t_prev; t_prev = t; }
  }
  if (dbg && tid == 0) { dbg[1] = tA; dbg[2] = tB; dbg[3] = tC; dbg[4] = t_prev - t_start; }

  // ---- write L (lower triangle, valid part) and the inverse block ------------------------------------
  {
    const int jp = 2 * (tid & 63), r0 = tid >> 6;
    const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    constexpr int NW = NT / 64, NIT = NB / NW;
#pragma unroll 8
    for (int it = 0; it < NIT; ++it) {
      const int i = r0 + NW * it;
      if (!FACTORED && i < nb && jp <= i) {
        const d2 lv = *reinterpret_cast<const d2*>(&S[i * LD + jp]);
        double* dst = A + (long)i * lda + jp;
        if (vec && jp + 1 <= i) *reinterpret_cast<d2*>(dst) = lv;
        else { dst[0] = lv.x; if (jp + 1 <= i) dst[1] = lv.y; }
      }
      d2 xv;
      auto xval = [&](int j) -> double {
        if (i < npad) {
          if (j < i) return S[j * LD + i];
          if (j == i) return xd[i];
          return 0.0;
        }
        return (j == i) ? 1.0 : 0.0;
      };
      xv.x = xval(jp);
      xv.y = xval(jp + 1);
      *reinterpret_cast<d2*>(&inv[i * NB + jp]) = xv;
    }
  }
  if (dbg && tid == 0) { dbg[5] = wall_clock64() - t_start; dbg[6] = clock64() - c_start; }
  if (!FACTORED && info) {
    // bad_col is wave-0 state; lane 0 of wave 0 reports (first failing pivot of the matrix wins)
    if (tid == 0 && bad_col >= 0 && info[blockIdx.x] == 0) info[blockIdx.x] = col0 + bad_col + 1;
  }
}

}  // namespace

static long long* g_leaf_dbg = nullptr;
extern "C" void gpk_debug_set_leaf_timing(long long* dev_buf) { g_leaf_dbg = dev_buf; }

int gpk_launch_leaf(hipStream_t s, double* A, long lda, long strideA, int nb, double* invd,
                    long strideInv, int* info, int col0, int batch, int already_factored) {
  if (nb <= 0 || nb > NB) return GPK_E_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    GPK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(leaf_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS));
    GPK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(leaf_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS));
    attr_set = true;
  }
  dim3 grid((unsigned)(batch > 0 ? batch : 1));
  if (already_factored)
    hipLaunchKernelGGL((leaf_kernel<true>), grid, dim3(NT), LEAF_LDS, s, A, lda, strideA, nb, invd,
                       strideInv, info, col0, g_leaf_dbg);
  else
    hipLaunchKernelGGL((leaf_kernel<false>), grid, dim3(NT), LEAF_LDS, s, A, lda, strideA, nb, invd,
                       strideInv, info, col0, g_leaf_dbg);
  GPK_LAUNCH_CHECK();
  return 0;
}
