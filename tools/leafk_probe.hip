// Probe for a "leaf + critical tiles" chain kernel (NEXT.md 1): what does one panel of the latency chain cost when the
// hand-offs leaf -> panel solve -> diagonal-tile update are IN-KERNEL flags between workgroups of one launch, and the
// bulk streams are released by hipStreamWaitValue32 on a flag the kernel writes?
//   workgroup 0      : stands in for the leaf (spins LEAF_US), then publishes a 128 x 128 operand with write-through
//                      (device-scope) stores and raises flagL
//   workgroups 1..H  : poll flagL, stage the operand into LDS with L1-bypassing LDS-DMA, one 16 x 128 x 128 MFMA sliver
//                      product each, publish the result (write-through), barrier over a counter, stage the 128 x 128
//                      block the first 8 helpers published, second sliver product, read-modify-write of a C sliver;
//                      the last helper raises flagK (signal memory, system scope)
//   stream B         : hipStreamWaitValue32(flagK >= p + 1) -> a one-lane kernel that stamps the clock
// Prints per-panel times of 16 back-to-back launches, the in-kernel hand-off latencies and the value-wait latency.
//   hipcc --offload-arch=gfx950 -O3 -o tools/leafk_probe tools/leafk_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorName(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = 128, LDK = NB + 2, NT = 512;
constexpr size_t LDS_BYTES = (size_t)(16 + NB) * LDK * sizeof(double);
constexpr long long SPIN_LIMIT = 200000;  // 2 ms of the 100 MHz clock: every wait is bounded

struct Args {
  double* inv;     // [16][128][128] operand published by workgroup 0 of launch p
  double* raw;     // [H*16][128] rows the helpers solve
  double* S;       // [16][H*16][128] published sliver results
  double* C;       // [H*16][128] read-modify-write target
  unsigned* flagL; unsigned* cnt; unsigned* flagK;   // flagK: signal memory
  long long* stamp;  // [16][8]: t_leaf_done, t_first_helper_saw, t_solve_done(last), t_barrier_passed(last), t_end(last), ...
  int p, H, leaf_ticks, bypass;
};

__device__ __forceinline__ bool wait_ge(const unsigned* f, unsigned v) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > SPIN_LIMIT) return false;
  }
  return true;
}

// rows [0, nrows) of src (row stride lds_) -> smem rows, 128 doubles each, one LDS-DMA instruction per row per wave
__device__ __forceinline__ void stage(const double* src, long ld, double* dst, int nrows, int bypass) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = wave; q < nrows; q += NT / 64) {
    const double* s = src + (long)q * ld + 2 * lane;
    if (bypass)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(dst + q * LDK), 16, 0, 16 /* sc1 */);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(dst + q * LDK), 16, 0, 0);
  }
}

__device__ __forceinline__ d4 sliver(const double* As, const double* Bs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const double* ap = As + r * LDK + g;
  const double* bp = Bs + (wave * 16 + r) * LDK + g;
  d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll 4
  for (int kk = 0; kk < NB / 4; kk += 2) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[kk * 4], bp[kk * 4], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[kk * 4 + 4], bp[kk * 4 + 4], a1, 0, 0, 0);
  }
  return a0 + a1;
}

__global__ __launch_bounds__(NT) void chain_probe(Args a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned epoch = (unsigned)a.p + 1u;
  long long* st = a.stamp + a.p * 8;
  if (blockIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < a.leaf_ticks) {}
    double* inv = a.inv + (long)a.p * NB * NB;
    for (int e = tid; e < NB * NB; e += NT) {
      const int i = e / NB, j = e % NB;
      const double v = (i == j) ? 1.0 : (j < i ? 1e-3 : 0.0);
      if (a.bypass) __hip_atomic_store(inv + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else inv[e] = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      if (!a.bypass) __threadfence();
      __hip_atomic_store(a.flagL, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      st[0] = wall_clock64();
    }
    return;
  }
  const int h = blockIdx.x - 1;
  double* As = smem;
  double* Bs = smem + 16 * LDK;
  __shared__ int ok;
  if (tid == 0) ok = wait_ge(a.flagL, epoch) ? 1 : 0;
  __syncthreads();
  if (!ok) return;
  if (!a.bypass) asm volatile("buffer_inv sc1" ::: "memory");
  if (h == 0 && tid == 0) st[1] = wall_clock64();
  // ---- solve: S_h = raw_h * inv^T ----
  stage(a.inv + (long)a.p * NB * NB, NB, Bs, NB, a.bypass);
  stage(a.raw + (long)h * 16 * NB, NB, As, 16, 0);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  d4 acc = sliver(As, Bs);
  const int r = lane & 15, g = lane >> 4;
  double* Sp = a.S + ((long)a.p * a.H * 16 + (long)h * 16) * NB;
  for (int e = 0; e < 4; ++e) {
    double* dst = Sp + (long)(g + 4 * e) * NB + wave * 16 + r;
    if (a.bypass) __hip_atomic_store(dst, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = acc[e];
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  // ---- barrier over the helpers ----
  if (tid == 0) {
    if (!a.bypass) __threadfence();
    const unsigned mine = __hip_atomic_fetch_add(a.cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (mine == (unsigned)a.H * epoch) st[2] = wall_clock64();
    ok = wait_ge(a.cnt, (unsigned)a.H * epoch) ? 1 : 0;
  }
  __syncthreads();
  if (!ok) return;
  if (!a.bypass) asm volatile("buffer_inv sc1" ::: "memory");
  // ---- diagonal-tile update: C_h -= S_h * S1^T  (S1 = the rows the first 8 helpers published) ----
  stage(a.S + (long)a.p * a.H * 16 * NB, NB, Bs, NB, a.bypass);
  stage(Sp, NB, As, 16, a.bypass);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  acc = sliver(As, Bs);
  double* Cp = a.C + (long)h * 16 * NB;
  for (int e = 0; e < 4; ++e) {
    double* dst = Cp + (long)(g + 4 * e) * NB + wave * 16 + r;
    *dst = *dst - acc[e];
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) {
    const unsigned mine = __hip_atomic_fetch_add(a.cnt + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (mine == (unsigned)a.H * epoch) {
      st[3] = wall_clock64();
      __hip_atomic_store(a.flagK, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void stamp_kernel(long long* out) { out[0] = wall_clock64(); }

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 16;
  const int leaf_us = argc > 2 ? atoi(argv[2]) : 35;
  const int bypass = argc > 3 ? atoi(argv[3]) : 1;
  const int with_wait = argc > 4 ? atoi(argv[4]) : 1;
  Args a{};
  a.H = H; a.leaf_ticks = leaf_us * 100; a.bypass = bypass;
  CK(hipMalloc((void**)&a.inv, sizeof(double) * 16 * NB * NB));
  CK(hipMalloc((void**)&a.raw, sizeof(double) * H * 16 * NB));
  CK(hipMalloc((void**)&a.S, sizeof(double) * 16 * H * 16 * NB));
  CK(hipMalloc((void**)&a.C, sizeof(double) * H * 16 * NB));
  CK(hipMalloc((void**)&a.flagL, 64));
  a.cnt = a.flagL + 4;
  CK(hipExtMallocWithFlags((void**)&a.flagK, 8, hipMallocSignalMemory));
  CK(hipMalloc((void**)&a.stamp, sizeof(long long) * 16 * 8 * 2));
  std::vector<double> hraw((size_t)H * 16 * NB);
  for (size_t i = 0; i < hraw.size(); ++i) hraw[i] = 1.0 + (double)(i % 7) * 0.25;
  CK(hipMemcpy(a.raw, hraw.data(), hraw.size() * sizeof(double), hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  hipStream_t P, B;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&P, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  long long* bstamp;
  CK(hipMalloc((void**)&bstamp, sizeof(long long) * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(a.flagL, 0, 64)); CK(hipMemset(a.flagK, 0, 8)); CK(hipMemset(a.stamp, 0, sizeof(long long) * 16 * 8));
    CK(hipMemset(a.C, 0, sizeof(double) * H * 16 * NB)); CK(hipMemset(bstamp, 0, sizeof(long long) * 16));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, P));
    for (int p = 0; p < 16; ++p) {
      a.p = p;
      hipLaunchKernelGGL(chain_probe, dim3(1 + H), dim3(NT), LDS_BYTES, P, a);
      if (with_wait) {
        CK(hipStreamWaitValue32(B, a.flagK, (unsigned)p + 1u, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, B, bstamp + p);
      }
    }
    CK(hipEventRecord(e1, P));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> st(16 * 8), bs(16);
    CK(hipMemcpy(st.data(), a.stamp, sizeof(long long) * 16 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(bs.data(), bstamp, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    std::vector<double> hc((size_t)H * 16 * NB);
    CK(hipMemcpy(hc.data(), a.C, hc.size() * sizeof(double), hipMemcpyDeviceToHost));
    double cs = 0;
    for (double v : hc) cs += v;
    double saw = 0, solve = 0, k1 = 0, vw = 0, period = 0;
    for (int p = 0; p < 16; ++p) {
      saw += (st[p * 8 + 1] - st[p * 8 + 0]) / 100.0;
      solve += (st[p * 8 + 2] - st[p * 8 + 0]) / 100.0;
      k1 += (st[p * 8 + 3] - st[p * 8 + 0]) / 100.0;
      vw += (bs[p] - st[p * 8 + 3]) / 100.0;
      if (p) period += (st[p * 8 + 0] - st[(p - 1) * 8 + 0]) / 100.0;
    }
    printf("H=%d leaf=%dus bypass=%d wait=%d rep=%d: 16 launches %.1f us (%.1f per panel; leaf-done to leaf-done %.1f) | flag seen +%.1f us, "
           "all slivers solved +%.1f, K1 done +%.1f (after the leaf) | value-wait kernel starts +%.1f after flagK | checksum %.6e\n",
           H, leaf_us, bypass, with_wait, rep, ms * 1e3, ms * 1e3 / 16, period / 15, saw / 16, solve / 16, k1 / 16, vw / 16, cs);
  }
  return 0;
}
