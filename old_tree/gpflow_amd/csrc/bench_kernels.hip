// Micro-benchmark kernels: fp64 MFMA issue rate (the denominator of every "fraction of peak" we
// quote is measured, not assumed) and a streaming fp64 store (HBM write ceiling for the K builder).
#include "gpk_internal.h"

namespace {
// 512-thread blocks: with <= 256 registers per lane hipcc keeps the accumulators in VGPRs; under
// __launch_bounds__(256) it parks them in AGPRs and copies all 64 of them through v_accvgpr_read/write on
// every iteration, which measures the copy loop (48 TFLOP/s) instead of the matrix pipe (77.4 TFLOP/s).
__global__ __launch_bounds__(512) void mfma_f64_rate_kernel(int iters, double* sink) {
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  if (s == 12345.678) sink[0] = s;  // never true; keeps the chain live
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sink[1] = (double)(t1 - t0) / ((double)iters * 8.0);  // shader cycles per MFMA (this wave)
    sink[2] = (double)(w1 - w0);                          // 100 MHz constant-clock ticks
    sink[3] = (double)(t1 - t0);
  }
}
__global__ __launch_bounds__(256) void stream_store_kernel(double* out, long n2) {
  d2* o = reinterpret_cast<d2*>(out);
  const d2 v = {1.0, 2.0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) o[i] = v;
}
}  // namespace

// flops issued = blocks * 8 waves * iters * 8 * 2048
extern "C" int gpk_bench_mfma_f64(void* stream, int blocks, int iters, double* sink) {
  hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, sink);
  GPK_LAUNCH_CHECK();
  return 0;
}
extern "C" int gpk_bench_stream_store(void* stream, double* out, long n_doubles) {
  hipLaunchKernelGGL(stream_store_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, out,
                     n_doubles / 2);
  GPK_LAUNCH_CHECK();
  return 0;
}
