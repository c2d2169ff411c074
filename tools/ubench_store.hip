// HBM write ceiling for fp64 outputs on MI355X: plain vs non-temporal 16-byte stores, 2 GiB, several grid shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NT>
__global__ __launch_bounds__(256) void k_stream(d2* o, long n2) {
  const d2 v = {1.0, 2.0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, &o[i]); else o[i] = v;
  }
}
// tile-shaped: each workgroup writes a 64 x 64 fp64 tile of an N x N row-major matrix (the K-builder's pattern)
template <int NT>
__global__ __launch_bounds__(256) void k_tile(double* o, int n) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
  const d2 v = {1.0, 2.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d2* p = reinterpret_cast<d2*>(o + (r0 + ty * 4 + i) * n + c0 + tx * 4);
    if (NT) { __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v, p + 1); } else { p[0] = v; p[1] = v; }
  }
}
// row-shaped: each wave writes one full 1 KiB row segment per instruction (64 lanes x 16 B), 128-column tiles
template <int NT>
__global__ __launch_bounds__(256) void k_rowtile(double* o, int n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 128;
  const d2 v = {1.0, 2.0};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d2* p = reinterpret_cast<d2*>(o + (r0 + wave * 8 + i) * n + c0 + lane * 2);
    if (NT) __builtin_nontemporal_store(v, p); else p[0] = v;
  }
}

template <typename F> static float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}
int main() {
  const int n = 16384; const long bytes = (long)n * n * 8;
  double* d; CK(hipMalloc(&d, bytes));
  for (int blocks : {2048, 8192, 65536}) {
    float a = timeit([&] { hipLaunchKernelGGL(k_stream<0>, dim3(blocks), dim3(256), 0, 0, (d2*)d, bytes / 16); });
    float b = timeit([&] { hipLaunchKernelGGL(k_stream<1>, dim3(blocks), dim3(256), 0, 0, (d2*)d, bytes / 16); });
    printf("{\"bench\": \"stream\", \"blocks\": %d, \"plain_TBps\": %.3f, \"nt_TBps\": %.3f}\n", blocks, bytes / a / 1e9, bytes / b / 1e9);
  }
  float a = timeit([&] { hipLaunchKernelGGL(k_tile<0>, dim3(n / 64, n / 64), dim3(256), 0, 0, d, n); });
  float b = timeit([&] { hipLaunchKernelGGL(k_tile<1>, dim3(n / 64, n / 64), dim3(256), 0, 0, d, n); });
  printf("{\"bench\": \"tile64x64\", \"plain_TBps\": %.3f, \"nt_TBps\": %.3f}\n", bytes / a / 1e9, bytes / b / 1e9);
  a = timeit([&] { hipLaunchKernelGGL(k_rowtile<0>, dim3(n / 128, n / 32), dim3(256), 0, 0, d, n); });
  b = timeit([&] { hipLaunchKernelGGL(k_rowtile<1>, dim3(n / 128, n / 32), dim3(256), 0, 0, d, n); });
  printf("{\"bench\": \"rowtile32x128\", \"plain_TBps\": %.3f, \"nt_TBps\": %.3f}\n", bytes / a / 1e9, bytes / b / 1e9);
  return 0;
}
