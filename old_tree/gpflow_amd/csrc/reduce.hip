// Small HBM-bound kernels around the factorisation: transposes/packing, row statistics of A^T,
// and the deterministic two-stage scalar reductions (ELBO data term, KL, log-dets, LML tail).
// All reductions are order-deterministic: stage 1 writes one partial per block, stage 2 (one block)
// sums them in a fixed order -- no floating-point atomics anywhere on this path.
#include "gpk_internal.h"

namespace {

constexpr int RB = 256;       // threads per reduction block
constexpr int MAXPART = 1024;  // max stage-1 blocks

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}
// valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sh[i];
  }
  return r;
}

// ---- stage 2: out = sum_t scale[t] * sum(part[t][0..count[t])) + add ------------------------------
struct FinalArgs {
  const double* part[4]; int count[4]; double scale[4]; int nterms; double add; double* out;
};
__global__ __launch_bounds__(RB) void final_sum_kernel(FinalArgs a) {
  __shared__ double sh[4];
  double total = a.add;
  for (int t = 0; t < a.nterms; ++t) {
    double v = 0.0;
    for (int i = threadIdx.x; i < a.count[t]; i += RB) v += a.part[t][i];
    const double r = block_sum(v, sh);
    total += a.scale[t] * r;
  }
  if (threadIdx.x == 0) *a.out = total;
}

// ---- zero the strict upper triangle -----------------------------------------------------------------
__global__ void zero_upper_kernel(double* A, int n, long lda, long strideA) {
  double* M = A + (long)blockIdx.z * strideA;
  const int r = blockIdx.y;
  for (int c = r + 1 + blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x)
    M[(long)r * lda + c] = 0.0;
}

// ---- transpose with optional triangular mask -------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const double* in, int rows, int cols,
                                                        long ldin, double* out, long ldout,
                                                        int mode, long stride_in, long stride_out) {
  __shared__ double tile[32][33];
  const double* I = in + (long)blockIdx.z * stride_in;
  double* O = out + (long)blockIdx.z * stride_out;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = by + k, c = bx + tx;
    double v = 0.0;
    if (r < rows && c < cols) {
      const bool keep = (mode == 0) || (mode == 1 && c <= r) || (mode == 2 && c >= r);
      if (keep) v = I[(long)r * ldin + c];
    }
    tile[k][tx] = v;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int r = bx + k, c = by + tx;  // out[r][c] = in[c][r]
    if (r < cols && c < rows) O[(long)r * ldout + c] = tile[tx][k];
  }
}

// out[c][r] = in[r][c] + shift  (used to lay (Y - mean)^T under the covariance matrix)
__global__ __launch_bounds__(256) void transpose_shift_kernel(const double* in, int rows, int cols,
                                                              long ldin, double* out, long ldout,
                                                              double shift) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = by + k, c = bx + tx;
    tile[k][tx] = (r < rows && c < cols) ? in[(long)r * ldin + c] + shift : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int r = bx + k, c = by + tx;
    if (r < cols && c < rows) out[(long)r * ldout + c] = tile[tx][k];
  }
}

// ---- row statistics of At [rows, m]:  sumsq[b], mv[b,p] = sum_k At[b,k] V[k,p],
//      wsq[p,b] = sum_k (At[b,k] W[k,p])^2.   One wave per row, 4 rows per block. --------------------
template <int PC>
__global__ __launch_bounds__(256) void row_stats_kernel(const double* __restrict__ At, int rows,
                                                        int m, long ldat,
                                                        const double* __restrict__ V,
                                                        const double* __restrict__ W, int P, int p0,
                                                        double alpha, double beta,
                                                        double* __restrict__ sumsq,
                                                        double* __restrict__ mv,
                                                        double* __restrict__ wsq) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  const double* a = At + (long)row * ldat;
  double s = 0.0, dv[PC], dw[PC];
#pragma unroll
  for (int q = 0; q < PC; ++q) { dv[q] = 0.0; dw[q] = 0.0; }
  for (int k = lane; k < m; k += 64) {
    const double x = a[k];
    s = fma(x, x, s);
#pragma unroll
    for (int q = 0; q < PC; ++q) {
      if (p0 + q < P) {
        if (V) dv[q] = fma(x, V[(long)k * P + p0 + q], dv[q]);
        if (W) { const double t = x * W[(long)k * P + p0 + q]; dw[q] = fma(t, t, dw[q]); }
      }
    }
  }
  s = wave_sum(s);
#pragma unroll
  for (int q = 0; q < PC; ++q) { dv[q] = wave_sum(dv[q]); dw[q] = wave_sum(dw[q]); }
  if (lane == 0) {
    if (sumsq && p0 == 0) sumsq[row] = (beta != 0.0 ? beta * sumsq[row] : 0.0) + alpha * s;
#pragma unroll
    for (int q = 0; q < PC; ++q)
      if (p0 + q < P) {
        if (V && mv) mv[(long)row * P + p0 + q] = dv[q];
        if (W && wsq) wsq[(long)(p0 + q) * rows + row] = dw[q];
      }
  }
}

// ---- the same for P separate At_p (SeparateIndependent latents): sumsq[p, b] = sum_k At_p[b,k]^2, mv[b, p] = sum_k At_p[b,k] V[k,p];
// one wave per (row, latent), blockIdx.y = p: ONE launch instead of P (plus P strided-column copies of V on the host side)
__global__ __launch_bounds__(256) void row_stats_sep_kernel(const double* __restrict__ At, long strideAt, int rows, int m, long ldat,
                                                            const double* __restrict__ V, int P, double* __restrict__ sumsq,
                                                            double* __restrict__ mv) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w, p = blockIdx.y;
  if (row >= rows) return;
  const double* a = At + (long)p * strideAt + (long)row * ldat;
  double s = 0.0, dv = 0.0;
  for (int k = lane; k < m; k += 64) {
    const double x = a[k];
    s = fma(x, x, s);
    dv = fma(x, V[(long)k * P + p], dv);
  }
  s = wave_sum(s);
  dv = wave_sum(dv);
  if (lane == 0) {
    sumsq[(long)p * rows + row] = s;
    mv[(long)row * P + p] = dv;
  }
}

// ---- out[i] = beta*out[i] + alpha * sum_j A[i,j] B[i,j]  (one wave per row) --------------------------
__global__ __launch_bounds__(256) void row_dot_kernel(const double* __restrict__ A, long lda,
                                                      const double* __restrict__ B, long ldb, int rows,
                                                      int cols, double alpha, double beta,
                                                      double* __restrict__ out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  const double* a = A + (long)row * lda;
  const double* b = B + (long)row * ldb;
  double s = 0.0;
  for (int k = lane; k < cols; k += 64) s = fma(a[k], b[k], s);
  s = wave_sum(s);
  if (lane == 0) out[row] = (beta != 0.0 ? beta * out[row] : 0.0) + alpha * s;
}

// ---- ssq[p,b] = sum_t part[p][t][b] -------------------------------------------------------------------
__global__ void sum_parts_kernel(const double* part, int nt, int rows, long stridePart, double* ssq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (b >= rows) return;
  const double* q = part + (long)p * stridePart;
  double s = 0.0;
  for (int t = 0; t < nt; ++t) s += q[(long)t * rows + b];
  ssq[(long)p * rows + b] = s;
}

// ---- out = alpha * sum_p part[p]  (optionally lower-triangular: zeros above the diagonal, diagonal scaled) --------
// The partial products of a split-K GEMM summed in a fixed order (p = 0, 1, ...): deterministic, one pass, 16-byte
// accesses; with lower != 0 entries above the diagonal are never read (a lower-only GEMM does not write those tiles).
// NP > 0: the number of parts is a compile-time constant and all NP loads of a thread are issued before the first add
// (with a run-time loop every add waited for its own load: 330 us for 8 parts of 2048^2, 0.4 TB/s); NP = 0: any count.
template <int NP>
__global__ __launch_bounds__(256) void combine_parts_kernel(const double* __restrict__ part, int np, long stridePart, int m,
                                                           int n, long ldp, double alpha, int lower, double diag_scale,
                                                           double* __restrict__ out, long ldo) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (c >= n) return;
  const bool two = c + 1 < n;
  const bool vec_ok = two && ((ldp & 1) == 0) && ((stridePart & 1) == 0) && ((reinterpret_cast<uintptr_t>(part) & 15) == 0);
  for (int r = blockIdx.y; r < m; r += gridDim.y) {
    const bool k0 = !lower || c <= r, k1 = two && (!lower || c + 1 <= r);
    double s0 = 0.0, s1 = 0.0;
    if (k0 || k1) {
      const double* q = part + (long)r * ldp + c;
      if (vec_ok && k0 && k1) {
        if (NP > 0) {
          d2 v[NP > 0 ? NP : 1];
#pragma unroll
          for (int p = 0; p < NP; ++p) v[p] = *reinterpret_cast<const d2*>(q + (long)p * stridePart);
#pragma unroll
          for (int p = 0; p < NP; ++p) { s0 += v[p].x; s1 += v[p].y; }
        } else {
          for (int p = 0; p < np; ++p, q += stridePart) {
            const d2 v = *reinterpret_cast<const d2*>(q);
            s0 += v.x; s1 += v.y;
          }
        }
      } else {
        for (int p = 0; p < np; ++p, q += stridePart) {
          if (k0) s0 += q[0];
          if (k1) s1 += q[1];
        }
      }
      s0 *= alpha; s1 *= alpha;
      if (lower) {
        if (c == r) s0 *= diag_scale;
        if (c + 1 == r) s1 *= diag_scale;
      }
    }
    double* o = out + (long)r * ldo + c;
    if (two && ((ldo & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
      *reinterpret_cast<d2*>(o) = (d2){k0 ? s0 : 0.0, k1 ? s1 : 0.0};
    } else {
      o[0] = k0 ? s0 : 0.0;
      if (two) o[1] = k1 ? s1 : 0.0;
    }
  }
}

// ---- Gaussian variational expectations, stage 1 -----------------------------------------------------
struct VarexpArgs {
  const double* Y; long ldy; const double* fmean; int rows, P;
  const double* s0; int s0_per_latent; const double* ssq;
  double knn[16]; int knn_per_latent;
  double noise, mean_const; double* fvar_out; double* part;
  const double* noise_rows;   // per-row noise variances [rows] (heteroskedastic Gaussian, scalar_continuous.py:92-111) or nullptr
  // (round 6) ssq given as the projection's column-slot partials [P][nt][rows] instead: summed here, slot 0 first (what sum_parts_kernel did)
  const double* ssq_part; int ssq_nt; long ssq_stride;
  // s0 / fmean come from a kernel on ANOTHER stream (row statistics beside the projection): wait for its word first (bounded)
  const int* wait_ptr; int wait_val; int* wait_info;
};
__global__ __launch_bounds__(RB) void varexp_kernel(VarexpArgs a) {
  __shared__ double sh[4];
  if (a.wait_ptr) {
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();   // 100 MHz
      while ((int)(__hip_atomic_load(a.wait_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.wait_val) < 0) {
        if (wall_clock64() - t0 >= 50000000LL) {
          if (a.wait_info) atomicMax(a.wait_info, 0x7fffffff);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (agent scope: the producer's end-of-kernel release made its stores visible)
  }
  const double log2pi = 1.8378770664093453;
  const double c0 = -0.5 * log2pi - 0.5 * log(a.noise);
  double acc = 0.0;
  const long total = (long)a.rows * a.P;
  for (long e = (long)blockIdx.x * RB + threadIdx.x; e < total; e += (long)gridDim.x * RB) {
    const int b = (int)(e / a.P), p = (int)(e - (long)b * a.P);
    double fv = a.knn[a.knn_per_latent ? p : 0];
    if (a.s0) fv -= a.s0_per_latent ? a.s0[(long)p * a.rows + b] : a.s0[b];
    if (a.ssq_part) {
      const double* q = a.ssq_part + (long)p * a.ssq_stride + b;
      double t = 0.0;
      for (int i = 0; i < a.ssq_nt; ++i) t += q[(long)i * a.rows];
      fv += t;
    } else if (a.ssq) fv += a.ssq[(long)p * a.rows + b];
    const double mu = a.fmean[e] + a.mean_const;
    const double dy = a.Y[(long)b * a.ldy + p] - mu;
    if (a.fvar_out) a.fvar_out[e] = fv;
    if (a.noise_rows) {   // (workgroup-uniform branch)
      const double nv = a.noise_rows[b];
      acc += (-0.5 * log2pi - 0.5 * log(nv)) - 0.5 * (dy * dy + fv) / nv;
    } else {
      acc += c0 - 0.5 * (dy * dy + fv) / a.noise;
    }
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) a.part[blockIdx.x] = r;
}

// ---- whitened KL, stage 1: sum q_mu^2 - sum log diag^2 + sum tril^2 -----------------------------------
__global__ __launch_bounds__(RB) void kl_white_kernel(const double* q_mu, const double* q_sqrt, int m,
                                                      int P, int q_diag, double* part) {
  __shared__ double sh[4];
  double acc = 0.0;
  const long nmu = (long)m * P;
  const long stride = (long)gridDim.x * RB, start = (long)blockIdx.x * RB + threadIdx.x;
  for (long e = start; e < nmu; e += stride) { const double v = q_mu[e]; acc += v * v; }
  if (q_diag) {
    for (long e = start; e < nmu; e += stride) {
      const double v = q_sqrt[e];
      acc += v * v - log(v * v);
    }
  } else {
    const long tot = (long)P * m * m;
    for (long e = start; e < tot; e += stride) {
      const long w = e % ((long)m * m);
      const int r = (int)(w / m), c = (int)(w - (long)r * m);
      if (c <= r) {
        const double v = q_sqrt[e];
        acc += v * v;
        if (c == r) acc -= log(v * v);
      }
    }
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// ---- sum log diag(L) per batch (one block per batch) ---------------------------------------------------
__global__ __launch_bounds__(RB) void sum_log_diag_kernel(const double* L, int n, long ldl,
                                                          long strideL, double* out) {
  __shared__ double sh[4];
  const double* M = L + (long)blockIdx.x * strideL;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += RB) acc += log(M[(long)i * ldl + i]);
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// out[b] = sum_i log(M_b[i,i]^2): the log-determinant of a covariance from a square root whose diagonal may carry either sign
// (kullback_leiblers.py:124: tf.math.log(tf.square(Lq_diag)))
__global__ __launch_bounds__(RB) void sum_log_diag_sq_kernel(const double* L, int n, long ldl, long strideL, double* out) {
  __shared__ double sh[4];
  const double* M = L + (long)blockIdx.x * strideL;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += RB) { const double v = M[(long)i * ldl + i]; acc += log(v * v); }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// ---- sum of squares of a matrix, stage 1 -----------------------------------------------------------------
__global__ __launch_bounds__(RB) void sumsq_kernel(const double* A, int rows, int cols, long lda,
                                                   int upper_only, double* part) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const double* a = A + (long)r * lda;
    for (int c = (upper_only ? r : 0) + threadIdx.x; c < cols; c += RB) acc = fma(a[c], a[c], acc);
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

int stage2(hipStream_t s, const FinalArgs& f) {
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(RB), 0, s, f);
  GPK_LAUNCH_CHECK();
  return 0;
}

int nblocks_for(long elems) {
  long b = (elems + RB * 4 - 1) / (RB * 4);
  if (b < 1) b = 1;
  if (b > MAXPART) b = MAXPART;
  return (int)b;
}

}  // namespace

// ================================================================================================
namespace {
__global__ __launch_bounds__(256) void set_identity_kernel(double* __restrict__ A, int n, long lda, long strideA) {
  const int row = blockIdx.y;
  double* a = A + (long)blockIdx.z * strideA;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) a[(long)row * lda + c] = (c == row) ? 1.0 : 0.0;
}
}  // namespace
namespace {
__global__ void noop_kernel() {}
}  // namespace
int gpk_launch_noop(hipStream_t s) {
  hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, s);
  GPK_LAUNCH_CHECK();
  return 0;
}

// ---- gate / signal kernels of the chain flags (potrf.hip, round 6) ---------------------------------------------------------
// hipStreamWaitValue32 / hipStreamWriteValue32 run as the runtime's own one-workgroup kernels behind queue packets: 5 - 7 us
// each between two kernels of a stream (rocprofv3: __amd_rocclr_streamOpsWait / Write).  A kernel of ours that follows another
// on its stream starts 0.3 us later.  So a stream that has to wait for a flag word enqueues this gate -- one wave, no LDS, one
// lane polling with s_sleep, bounded like the in-kernel waits of the GEMM kernels (0.5 s, then the status word becomes
// INT_MAX) -- and a stream that has to publish one enqueues the one-thread store.  The end-of-kernel release of whatever ran
// before the store / the acquire at the start of whatever follows the gate order the data as the packets did.
namespace {
__global__ void wait_flag_kernel(const int* __restrict__ ptr, int val, int* __restrict__ info) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();   // 100 MHz
    while ((int)(__hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - val) < 0) {
      if (wall_clock64() - t0 >= 50000000LL) {
        if (info) atomicMax(info, 0x7fffffff);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
__global__ void set_flag_kernel(int* __restrict__ ptr, int val) {
  __hip_atomic_store(ptr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
int gpk_launch_wait_flag(hipStream_t s, const int* ptr, int val, int* info) {
  hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, s, ptr, val, info);
  GPK_LAUNCH_CHECK();
  return 0;
}
int gpk_launch_set_flag(hipStream_t s, int* ptr, int val) {
  hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, s, ptr, val);
  GPK_LAUNCH_CHECK();
  return 0;
}

// ---- can two kernels of this process run at the same time? -----------------------------------------------------------------
// The chain flags of potrf.hip let a kernel wait in-kernel for a word that a kernel (or stream write) on ANOTHER stream sets.
// Under a tool that serialises kernel execution (rocprofv3 --pmc, AMD_SERIALIZE_KERNEL) the producer would never start while the
// consumer spins: a deadlock inside the runtime's own stream-wait kernel, which has no timeout.  So the first factorisation of a
// device asks: a kernel that waits at most 2 ms for a word, and one on a second stream that sets it.
__global__ void probe_wait_kernel(const int* flag, int* result) {
  const long long t0 = wall_clock64();   // 100 MHz
  int seen = 0;
  while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) && wall_clock64() - t0 < 200000LL)
    __builtin_amdgcn_s_sleep(8);
  *result = seen ? 1 : 0;
}
__global__ void probe_set_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
int gpk_probe_concurrent_kernels(hipStream_t a, hipStream_t b, int* scratch /* 2 device ints */, int* concurrent) {
  GPK_HIP(hipMemsetAsync(scratch, 0, 2 * sizeof(int), a));
  GPK_HIP(hipStreamSynchronize(a));
  hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, a, scratch, scratch + 1);
  GPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, b, scratch);
  GPK_LAUNCH_CHECK();
  GPK_HIP(hipStreamSynchronize(a));
  GPK_HIP(hipStreamSynchronize(b));
  int h[2] = {0, 0};
  GPK_HIP(hipMemcpy(h, scratch, sizeof(h), hipMemcpyDeviceToHost));
  *concurrent = h[1];
  GPK_HIP(hipMemset(scratch, 0, 2 * sizeof(int)));
  return 0;
}

// A[i,i] += v[i]:  add_noise_cov with a per-row likelihood variance (utilities/model_utils.py:33-38, 46-50)
__global__ __launch_bounds__(256) void diag_add_kernel(double* __restrict__ A, int n, long lda, const double* __restrict__ v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) A[(long)i * lda + i] += v[i];
}
__global__ __launch_bounds__(256) void diag_add_scalar_kernel(double* __restrict__ A, int n, long lda, double v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) A[(long)i * lda + i] += v;
}
int gpk_launch_diag_add_scalar(hipStream_t s, double* A, int n, long lda, double v) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(diag_add_scalar_kernel, dim3((n + 255) / 256), dim3(256), 0, s, A, n, lda, v);
  GPK_LAUNCH_CHECK();
  return 0;
}
extern "C" int gpk_diag_add(void* stream, double* A, int n, long lda, const double* v) {
  if (!A || !v || n < 0 || lda < n) return GPK_E_ARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(diag_add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, n, lda, v);
  GPK_LAUNCH_CHECK();
  return 0;
}
int gpk_launch_set_identity(hipStream_t s, double* A, int n, long lda, int batch, long strideA) {
  if (n <= 0) return 0;
  dim3 grid((unsigned)gpk_cdiv(n, 256), (unsigned)n, (unsigned)(batch > 0 ? batch : 1));
  hipLaunchKernelGGL(set_identity_kernel, grid, dim3(256), 0, s, A, n, lda, strideA);
  GPK_LAUNCH_CHECK();
  return 0;
}

int gpk_launch_zero_upper(hipStream_t s, double* A, int n, long lda, int batch, long strideA) {
  if (n <= 1) return 0;
  int gx = gpk_cdiv(n, 256);
  if (gx > 16) gx = 16;
  dim3 grid((unsigned)gx, (unsigned)n, (unsigned)(batch > 0 ? batch : 1));
  hipLaunchKernelGGL(zero_upper_kernel, grid, dim3(256), 0, s, A, n, lda, strideA);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t gpk_reduce_workspace_bytes(int n) {
  (void)n;
  return (size_t)4 * MAXPART * sizeof(double);
}

extern "C" int gpk_transpose(void* stream, const double* in, int rows, int cols, long ldin,
                             double* out, long ldout, int mode, int batch, long stride_in,
                             long stride_out) {
  if (!in || !out || rows < 0 || cols < 0) return GPK_E_ARG;
  if (rows == 0 || cols == 0) return 0;
  dim3 grid((unsigned)gpk_cdiv(cols, 32), (unsigned)gpk_cdiv(rows, 32),
            (unsigned)(batch > 0 ? batch : 1));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, rows, cols, ldin,
                     out, ldout, mode, stride_in, stride_out);
  GPK_LAUNCH_CHECK();
  return 0;
}

// sumsq[b] = beta*sumsq[b] + alpha*sum_k At^2 ; mv = At V ; wsq[p,b] = sum_k (At W[:,p])^2
extern "C" int gpk_row_stats(void* stream, const double* At, int rows, int m, long ldat,
                             const double* V, const double* W, int P, double alpha, double beta,
                             double* sumsq, double* mv, double* wsq) {
  if (!At || rows < 0 || m < 0) return GPK_E_ARG;
  if (rows == 0) return 0;
  const int np = (V || W) ? P : 0;
  const dim3 grid((unsigned)gpk_cdiv(rows, 4));
  int p0 = 0;
  do {
    hipLaunchKernelGGL((row_stats_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, At, rows, m,
                       ldat, V, W, np, p0, alpha, beta, sumsq, mv, wsq);
    GPK_LAUNCH_CHECK();
    p0 += 4;
  } while (p0 < np);
  return 0;
}

int gpk_launch_row_stats_sep(hipStream_t s, const double* At, long strideAt, int rows, int m, long ldat, const double* V, int P,
                             double* sumsq, double* mv) {
  if (!At || !V || !sumsq || !mv || rows < 0 || m < 0 || P <= 0) return GPK_E_ARG;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(row_stats_sep_kernel, dim3((unsigned)gpk_cdiv(rows, 4), (unsigned)P), dim3(256), 0, s, At, strideAt, rows, m,
                     ldat, V, P, sumsq, mv);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_row_dot(void* stream, const double* A, long lda, const double* B, long ldb,
                           int rows, int cols, double alpha, double beta, double* out) {
  if (!A || !B || !out || rows < 0 || cols < 0) return GPK_E_ARG;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(row_dot_kernel, dim3((unsigned)gpk_cdiv(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, A, lda, B, ldb, rows, cols, alpha, beta, out);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_row_sumsq(void* stream, const double* A, int rows, int cols, long lda,
                             double alpha, double beta, double* out) {
  return gpk_row_stats(stream, A, rows, cols, lda, nullptr, nullptr, 0, alpha, beta, out, nullptr,
                       nullptr);
}

int gpk_launch_sum_parts(hipStream_t s, const double* part, int nt, int rows, long stridePart, int P,
                         double* ssq) {
  if (rows == 0 || P == 0) return 0;
  dim3 grid((unsigned)gpk_cdiv(rows, 256), (unsigned)P);
  hipLaunchKernelGGL(sum_parts_kernel, grid, dim3(256), 0, s, part, nt, rows, stridePart, ssq);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_combine_parts(void* stream, const double* parts, int nparts, long stride_part, int m, int n, long ldp,
                                 double alpha, int lower, double diag_scale, double* out, long ldo) {
  if (!parts || !out || nparts <= 0 || m < 0 || n < 0 || ldp < n || ldo < n) return GPK_E_ARG;
  if (m == 0 || n == 0) return 0;
  dim3 grid((unsigned)gpk_cdiv(gpk_cdiv(n, 2), 256), (unsigned)(m < 65535 ? m : 65535));
#define GPK_COMBINE(NP)                                                                                               \
  hipLaunchKernelGGL((combine_parts_kernel<NP>), grid, dim3(256), 0, (hipStream_t)stream, parts, nparts, stride_part, m, n, \
                     ldp, alpha, lower, diag_scale, out, ldo)
  switch (nparts) {
    case 1: GPK_COMBINE(1); break;
    case 2: GPK_COMBINE(2); break;
    case 4: GPK_COMBINE(4); break;
    case 8: GPK_COMBINE(8); break;
    case 16: GPK_COMBINE(16); break;
    case 32: GPK_COMBINE(32); break;
    default: GPK_COMBINE(0); break;
  }
#undef GPK_COMBINE
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_gaussian_varexp_sum(void* stream, const double* Y, long ldy, const double* fmean,
                                       int rows, int P, const double* s0, int s0_per_latent,
                                       const double* ssq, const double* knn_host,
                                       int knn_per_latent, double noise_variance, const double* noise_rows,
                                       double mean_const, double* fvar_out, double* out, void* ws, size_t ws_bytes) {
  if (!Y || !fmean || !knn_host || !out || P <= 0 || P > 16 || rows < 0) return GPK_E_ARG;
  if (!ws || ws_bytes < gpk_reduce_workspace_bytes(rows)) return GPK_E_WORKSPACE;
  VarexpArgs a{};
  a.Y = Y; a.ldy = ldy; a.fmean = fmean; a.rows = rows; a.P = P;
  a.s0 = s0; a.s0_per_latent = s0_per_latent; a.ssq = ssq;
  for (int i = 0; i < (knn_per_latent ? P : 1); ++i) a.knn[i] = knn_host[i];
  a.knn_per_latent = knn_per_latent; a.noise = noise_variance; a.mean_const = mean_const;
  a.fvar_out = fvar_out; a.part = (double*)ws; a.noise_rows = noise_rows;
  const int nb = nblocks_for((long)rows * P);
  hipLaunchKernelGGL(varexp_kernel, dim3(nb), dim3(RB), 0, (hipStream_t)stream, a);
  GPK_LAUNCH_CHECK();
  FinalArgs f{};
  f.nterms = 1; f.part[0] = a.part; f.count[0] = nb; f.scale[0] = 1.0; f.add = 0.0; f.out = out;
  return stage2((hipStream_t)stream, f);
}

extern "C" int gpk_gauss_kl_white(void* stream, const double* q_mu, const double* q_sqrt, int m,
                                  int P, int q_diag, double* out, void* ws, size_t ws_bytes) {
  if (!q_mu || !q_sqrt || !out || m <= 0 || P <= 0) return GPK_E_ARG;
  if (!ws || ws_bytes < gpk_reduce_workspace_bytes(m)) return GPK_E_WORKSPACE;
  const long elems = q_diag ? (long)m * P : (long)P * m * m;
  const int nb = nblocks_for(elems);
  double* part = (double*)ws;
  hipLaunchKernelGGL(kl_white_kernel, dim3(nb), dim3(RB), 0, (hipStream_t)stream, q_mu, q_sqrt, m, P,
                     q_diag, part);
  GPK_LAUNCH_CHECK();
  FinalArgs f{};
  f.nterms = 1; f.part[0] = part; f.count[0] = nb; f.scale[0] = 0.5;
  f.add = -0.5 * (double)m * (double)P; f.out = out;
  return stage2((hipStream_t)stream, f);
}

extern "C" int gpk_sum_log_diag(void* stream, const double* L, int n, long ldl, int batch,
                                long strideL, double* out) {
  if (!L || !out || n <= 0) return GPK_E_ARG;
  hipLaunchKernelGGL(sum_log_diag_kernel, dim3((unsigned)(batch > 0 ? batch : 1)), dim3(RB), 0,
                     (hipStream_t)stream, L, n, ldl, strideL, out);
  GPK_LAUNCH_CHECK();
  return 0;
}

// Un-whitened KL with a diagonal q_sqrt (kullback_leiblers.py:128-165, q_diag and K given): per inducing point i
//   part += (K^-1)_ii * sum_p w_ip^2 - sum_p log(w_ip^2),   (K^-1)_ii = |row i of L^-T|^2  (rows of LinvT, upper triangular)
__global__ __launch_bounds__(RB) void kl_unwhite_diag_kernel(const double* __restrict__ LinvT, long ldl, int m,
                                                            const double* __restrict__ W, int P, double* __restrict__ part) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = blockIdx.x; i < m; i += gridDim.x) {
    double ss = 0.0;
    for (int k = i + threadIdx.x; k < m; k += RB) {
      const double v = LinvT[(long)i * ldl + k];
      ss += v * v;
    }
    const double kinv = block_sum(ss, sh);
    __syncthreads();
    if (threadIdx.x == 0) {
      double w2 = 0.0, lg = 0.0;
      for (int p = 0; p < P; ++p) {
        const double w = W[(long)i * P + p];
        w2 += w * w;
        lg += log(w * w);
      }
      acc += kinv * w2 - lg;
    }
  }
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
int gpk_launch_kl_unwhite_diag_stage1(hipStream_t s, const double* LinvT, long ldl, int m, const double* W, int P, double* part,
                                      int* count) {
  const int nb = m < GPK_REDUCE_MAXPART ? m : GPK_REDUCE_MAXPART;
  hipLaunchKernelGGL(kl_unwhite_diag_kernel, dim3(nb), dim3(RB), 0, s, LinvT, ldl, m, W, P, part);
  GPK_LAUNCH_CHECK();
  *count = nb;
  return 0;
}

int gpk_launch_sum_log_diag_sq(hipStream_t s, const double* L, int n, long ldl, int batch, long strideL, double* out) {
  if (!L || !out || n <= 0) return GPK_E_ARG;
  hipLaunchKernelGGL(sum_log_diag_sq_kernel, dim3((unsigned)(batch > 0 ? batch : 1)), dim3(RB), 0, s, L, n, ldl, strideL, out);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_sumsq(void* stream, const double* A, int rows, int cols, long lda, int upper_only,
                         double* out, void* ws, size_t ws_bytes) {
  if (!A || !out || rows < 0 || cols < 0) return GPK_E_ARG;
  if (!ws || ws_bytes < gpk_reduce_workspace_bytes(rows)) return GPK_E_WORKSPACE;
  int nb = rows < MAXPART ? rows : MAXPART;
  if (nb < 1) nb = 1;
  double* part = (double*)ws;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(RB), 0, (hipStream_t)stream, A, rows, cols, lda,
                     upper_only, part);
  GPK_LAUNCH_CHECK();
  FinalArgs f{};
  f.nterms = 1; f.part[0] = part; f.count[0] = nb; f.scale[0] = 1.0; f.add = 0.0; f.out = out;
  return stage2((hipStream_t)stream, f);
}

// out = sum_t scale[t]*sum(part[t][0..count[t])) + add   (used by the fused drivers)
int gpk_launch_final(hipStream_t s, int nterms, const double* const* part, const int* count,
                     const double* scale, double add, double* out) {
  FinalArgs f{};
  f.nterms = nterms;
  for (int t = 0; t < nterms; ++t) { f.part[t] = part[t]; f.count[t] = count[t]; f.scale[t] = scale[t]; }
  f.add = add; f.out = out;
  return stage2(s, f);
}

// stage-1 launchers reused by the fused drivers (partials land in `part`, count returned)
int gpk_launch_sumsq_stage1(hipStream_t s, const double* A, int rows, int cols, long lda,
                            int upper_only, double* part, int* count) {
  int nb = rows < MAXPART ? rows : MAXPART;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(RB), 0, s, A, rows, cols, lda, upper_only, part);
  GPK_LAUNCH_CHECK();
  *count = nb;
  return 0;
}
int gpk_launch_varexp_stage1(hipStream_t s, const double* Y, long ldy, const double* fmean, int rows,
                             int P, const double* s0, int s0_per_latent, const double* ssq,
                             const double* knn_host, int knn_per_latent, double noise,
                             double mean_const, double* fvar_out, double* part, int* count, const double* noise_rows,
                             const VarexpExtra* ex) {
  VarexpArgs a{};
  if (ex) {
    a.ssq_part = ex->ssq_part; a.ssq_nt = ex->ssq_nt; a.ssq_stride = ex->ssq_stride;
    a.wait_ptr = ex->wait_ptr; a.wait_val = ex->wait_val; a.wait_info = ex->wait_info;
  }
  a.Y = Y; a.ldy = ldy; a.fmean = fmean; a.rows = rows; a.P = P;
  a.s0 = s0; a.s0_per_latent = s0_per_latent; a.ssq = ssq;
  for (int i = 0; i < (knn_per_latent ? P : 1); ++i) a.knn[i] = knn_host[i];
  a.knn_per_latent = knn_per_latent; a.noise = noise; a.mean_const = mean_const;
  a.fvar_out = fvar_out; a.part = part; a.noise_rows = noise_rows;
  const int nb = nblocks_for((long)rows * P);
  hipLaunchKernelGGL(varexp_kernel, dim3(nb), dim3(RB), 0, s, a);
  GPK_LAUNCH_CHECK();
  *count = nb;
  return 0;
}
int gpk_launch_kl_white_stage1(hipStream_t s, const double* q_mu, const double* q_sqrt, int m, int P,
                               int q_diag, double* part, int* count) {
  const long elems = q_diag ? (long)m * P : (long)P * m * m;
  const int nb = nblocks_for(elems);
  hipLaunchKernelGGL(kl_white_kernel, dim3(nb), dim3(RB), 0, s, q_mu, q_sqrt, m, P, q_diag, part);
  GPK_LAUNCH_CHECK();
  *count = nb;
  return 0;
}
int gpk_launch_transpose_shift(hipStream_t s, const double* in, int rows, int cols, long ldin,
                               double* out, long ldout, double shift) {
  if (rows == 0 || cols == 0) return 0;
  dim3 grid((unsigned)gpk_cdiv(cols, 32), (unsigned)gpk_cdiv(rows, 32), 1);
  hipLaunchKernelGGL(transpose_shift_kernel, grid, dim3(256), 0, s, in, rows, cols, ldin, out, ldout,
                     shift);
  GPK_LAUNCH_CHECK();
  return 0;
}

// ---- result mailbox: device scalars -> mapped host memory, sequence word last (gpk.h) --------------------------------
namespace {
__global__ void publish_host_kernel(const double* __restrict__ src, int n, const int* __restrict__ info, double* vals, int* tail,
                                    int seq) {
  if (threadIdx.x == 0) {
    for (int i = 0; i < n; ++i) __hip_atomic_store(vals + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(tail, info ? info[0] : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(tail + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // after everything above
  }
}
}  // namespace

extern "C" int gpk_publish_host(void* stream, const double* src, int n, const int* info, void* host_dst, int seq) {
  if (!src || !host_dst || n <= 0 || n > 16) return GPK_E_ARG;
  double* vals = (double*)host_dst;
  hipLaunchKernelGGL(publish_host_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, n, info, vals, (int*)(vals + n), seq);
  GPK_LAUNCH_CHECK();
  return 0;
}

// ---- glue of the reverse pass as single launches (round 6, late) -------------------------------------------------------------
// The tail of a training step was ~70 torch elementwise / reduction launches of 4 - 5 us each on arrays of a few thousand
// elements (profiles/r06_train_timeline_gated_side_branch.txt: 0.44 ms behind the last GEMM).  Three kernels replace most of
// them: the moment rows [1; B^T; (B^T)^2] of a stationary kernel's adjoint, the adjoint's tail (input gradient, lengthscale and
// variance gradients from G [1, B, B^2]) and one Adam update per variable.
namespace {
__global__ __launch_bounds__(256) void moment_rows_kernel(const double* __restrict__ B, long ldb, int n2, int d, double* __restrict__ Vt,
                                                          long ldv) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n2) return;
  Vt[j] = 1.0;
  for (int c = 0; c < d; ++c) {
    const double b = B[(long)j * ldb + c];
    Vt[(long)(1 + c) * ldv + j] = b;
    Vt[(long)(1 + d + c) * ldv + j] = b * b;
  }
}

constexpr int AT_THREADS = 1024;
// one workgroup; thread t owns input dimension t % d and walks the rows t / d, t / d + rpp, ... (rpp = AT_THREADS / d rows per pass):
// every partial sum has a fixed set of terms in a fixed order, and the partials meet in LDS in thread order -- deterministic
__global__ __launch_bounds__(AT_THREADS) void adjoint_tail_kernel(const double* __restrict__ R, long ldr, const double* __restrict__ A,
                                                                  long lda, int n1, int d, const double* __restrict__ ls, double variance,
                                                                  int symmetric, const double* __restrict__ sum_kbar_k,
                                                                  double* __restrict__ Abar, long ldab, double* __restrict__ small,
                                                                  int accumulate, double dvar_add) {
  __shared__ double sh[AT_THREADS];
  __shared__ double sh_rs[AT_THREADS];
  const int t = threadIdx.x;
  const int rpp = AT_THREADS / d;
  const int c = t % d, r0 = t / d;
  double acc = 0.0, acc_rs = 0.0;
  if (r0 < rpp) {
    const double l = ls[c];
    const double il2 = 1.0 / (l * l);
    for (int i = r0; i < n1; i += rpp) {
      const double* Ri = R + (long)i * ldr;
      const double rs = Ri[0], gb = Ri[1 + c], gb2 = Ri[1 + d + c];
      const double a = A[(long)i * lda + c];
      const double T = gb - a * rs;
      double ab;
      if (symmetric) {
        ab = 2.0 * T * il2;
        acc += a * ab;
      } else {
        ab = T * il2;
        acc += gb2 - a * (gb + T);
      }
      double* o = Abar + (long)i * ldab + c;
      *o = accumulate ? *o + ab : ab;
      if (c == 0) acc_rs += rs;
    }
  }
  sh[t] = acc;
  sh_rs[t] = acc_rs;
  __syncthreads();
  if (t < d) {   // (r0 == 0: this thread's own column)
    double s = 0.0;
    for (int q = 0; q < rpp; ++q) s += sh[q * d + t];
    const double l = ls[t];
    const double r = symmetric ? -s / l : s / (l * l * l);
    small[1 + t] = accumulate ? small[1 + t] + r : r;
  }
  if (t == 0) {
    double s = 0.0;
    if (sum_kbar_k) s = sum_kbar_k[0];
    else
      for (int q = 0; q < rpp; ++q) s += sh_rs[q * d];
    const double r = s / variance + dvar_add;
    small[0] = accumulate ? small[0] + r : r;
  }
}

// tf.keras Adam on one variable, minimising -F:  g is dF/dp  (m, v, p updated in place; step = lr sqrt(1 - b2^t) / (1 - b1^t) from the host)
__global__ __launch_bounds__(256) void adam_kernel(double* __restrict__ p, const double* __restrict__ g, double* __restrict__ m,
                                                   double* __restrict__ v, long n, double b1, double b2, double eps, double step,
                                                   double gsign) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double gi = gsign * g[i];
    const double mi = b1 * m[i] + (1.0 - b1) * gi;
    const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step * mi / (sqrt(vi) + eps);
  }
}

// out = alpha X + U V^T for a thin U [m, k], V [n, k], k <= 16: the start of At_bar = r q_mu^T - 2 c P At + ... (one pass over X
// instead of a K = k GEMM plus an axpy pass)
__global__ __launch_bounds__(256) void lowrank_axpy_kernel(double alpha, const double* __restrict__ X, long ldx, const double* __restrict__ U,
                                                           long ldu, const double* __restrict__ V, long ldv, int m, int n, int k,
                                                           double* __restrict__ out, long ldo) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (c >= n) return;
  const bool two = c + 1 < n;
  double v0[16], v1[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    v0[q] = q < k ? V[(long)c * ldv + q] : 0.0;
    v1[q] = (q < k && two) ? V[(long)(c + 1) * ldv + q] : 0.0;
  }
  const bool vec = two && !(ldx & 1) && !(ldo & 1) && !(reinterpret_cast<uintptr_t>(X) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15);
  for (int r = blockIdx.y; r < m; r += gridDim.y) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (q < k) {
        const double u = U[(long)r * ldu + q];
        s0 += u * v0[q];
        s1 += u * v1[q];
      }
    const double* x = X + (long)r * ldx + c;
    double* o = out + (long)r * ldo + c;
    if (vec) {
      const d2 xv = *reinterpret_cast<const d2*>(x);
      *reinterpret_cast<d2*>(o) = (d2){alpha * xv.x + s0, alpha * xv.y + s1};
    } else {
      o[0] = alpha * x[0] + s0;
      if (two) o[1] = alpha * x[1] + s1;
    }
  }
}

// out = (S + S^T) / 2 of a square matrix, in place: 32 x 32 tile pairs (bi >= bj), both tiles through LDS
__global__ __launch_bounds__(256) void symmetrize_kernel(double* __restrict__ S, int n, long lds) {
  __shared__ double ta[32][33], tb[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    ta[r][tx] = (i < n && j < n) ? S[(long)i * lds + j] : 0.0;        // S[bi-block, bj-block]
    const int i2 = bj * 32 + r, j2 = bi * 32 + tx;
    tb[r][tx] = (i2 < n && j2 < n) ? S[(long)i2 * lds + j2] : 0.0;    // S[bj-block, bi-block]
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    if (i < n && j < n) S[(long)i * lds + j] = 0.5 * (ta[r][tx] + tb[tx][r]);
    if (bi != bj) {
      const int i2 = bj * 32 + r, j2 = bi * 32 + tx;
      if (i2 < n && j2 < n) S[(long)i2 * lds + j2] = 0.5 * (tb[r][tx] + ta[tx][r]);
    }
  }
}
}  // namespace

extern "C" int gpk_moment_rows(void* stream, const double* B, long ldb, int n2, int d, double* Vt, long ldv) {
  if (!B || !Vt || n2 < 0 || d <= 0 || ldb < d || ldv < n2) return GPK_E_ARG;
  if (n2 == 0) return 0;
  hipLaunchKernelGGL(moment_rows_kernel, dim3((unsigned)gpk_cdiv(n2, 256)), dim3(256), 0, (hipStream_t)stream, B, ldb, n2, d, Vt, ldv);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_stationary_adjoint_tail(void* stream, const double* R, long ldr, const double* A, long lda, int n1, int d,
                                           const double* ls_dev, double variance, int symmetric, const double* sum_kbar_k,
                                           double* Abar, long ldab, double* small, int accumulate, double dvar_add) {
  if (!R || !A || !ls_dev || !Abar || !small || n1 < 0 || d <= 0 || d > AT_THREADS || ldr < 1 + 2 * d || lda < d || ldab < d)
    return GPK_E_ARG;
  hipLaunchKernelGGL(adjoint_tail_kernel, dim3(1), dim3(AT_THREADS), 0, (hipStream_t)stream, R, ldr, A, lda, n1, d, ls_dev, variance,
                     symmetric, sum_kbar_k, Abar, ldab, small, accumulate, dvar_add);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_adam_step(void* stream, double* p, const double* g, double* m, double* v, long n, double beta1, double beta2,
                             double epsilon, double step, int maximise) {
  if (!p || !g || !m || !v || n < 0) return GPK_E_ARG;
  if (n == 0) return 0;
  const long nb = (n + 255) / 256;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2,
                     epsilon, step, maximise ? -1.0 : 1.0);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_symmetrize(void* stream, double* S, int n, long lds) {
  if (!S || n < 0 || lds < n) return GPK_E_ARG;
  if (n == 0) return 0;
  const unsigned nb = (unsigned)gpk_cdiv(n, 32);
  hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, (hipStream_t)stream, S, n, lds);
  GPK_LAUNCH_CHECK();
  return 0;
}

extern "C" int gpk_lowrank_axpy(void* stream, double alpha, const double* X, long ldx, const double* U, long ldu, const double* V, long ldv,
                                int m, int n, int k, double* out, long ldo) {
  if (!X || !U || !V || !out || m < 0 || n < 0 || k <= 0 || k > 16 || ldx < n || ldo < n || ldu < k || ldv < k) return GPK_E_ARG;
  if (m == 0 || n == 0) return 0;
  dim3 grid((unsigned)gpk_cdiv(gpk_cdiv(n, 2), 256), (unsigned)(m < 2048 ? m : 2048));
  hipLaunchKernelGGL(lowrank_axpy_kernel, grid, dim3(256), 0, (hipStream_t)stream, alpha, X, ldx, U, ldu, V, ldv, m, n, k, out, ldo);
  GPK_LAUNCH_CHECK();
  return 0;
}
