// fp64 issue-rate micro-benchmarks for gfx950 (standalone; `make -C tools` builds tools/ubench_f64).
// The in-image hardware guide lists no fp64 peak, so every "fraction of peak" in this repo is quoted
// against the datasheet 78.6 TFLOP/s AND against what these loops measure on the chip:
//   mode 0  v_mfma_f64_16x16x4_f64   (NACC independent accumulators per wave)
//   mode 1  v_mfma_f64_4x4x4_4b_f64
//   mode 2  v_fma_f64 (VALU), NACC independent chains per lane
//   mode 3  mixed: waves 0..3 of a 512-thread block issue MFMA, waves 4..7 issue VALU FMA (same SIMDs)
// Prints one JSON line per configuration: TFLOP/s (HIP events), shader cycles per instruction
// (s_memtime), effective shader clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double lane_val(unsigned seed) {
  unsigned h = (threadIdx.x + 1u) * 2654435761u ^ (seed * 40503u + blockIdx.x * 97u);
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return ((double)(h & 0xFFFFFF) / 16777216.0 - 0.5) * 1.9;  // (-0.95, 0.95), full mantissa toggling
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(int iters, double* out) {
  const double x = lane_val(1), y = lane_val(2);
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678) out[8] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(w1 - w0); }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(int iters, double* out) {
  const double x = lane_val(1), y = lane_val(2);
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678) out[8] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(w1 - w0); }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_valu(int iters, double* out) {
  const double x = lane_val(1) * 0.5, y = lane_val(2);
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = lane_val(3 + i);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], x, y);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678) out[8] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(w1 - w0); }
}

// mixed: MFMA waves and VALU waves co-resident on every SIMD; vratio VALU fmas issued per MFMA wave-iteration
template <int NACC>
__global__ __launch_bounds__(512) void k_mixed(int iters, int viters, double* out) {
  const int wave = threadIdx.x >> 6;
  const double x = lane_val(1) * 0.5, y = lane_val(2);
  const long long t0 = clock64(), w0 = wall_clock64();
  double s = 0.0;
  if (wave < 4) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  } else {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = lane_val(3 + i);
    for (int it = 0; it < viters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], x, y);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  if (s == 12345.678) out[8] = s;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) {
    out[wave < 4 ? 0 : 2] = (double)(t1 - t0);
    out[wave < 4 ? 1 : 3] = (double)(w1 - w0);
  }
}

// same-wave interleave: each wave issues NACC MFMAs and VPER VALU fmas per iteration
template <int NACC, int VPER>
__global__ __launch_bounds__(256) void k_inter(int iters, double* out) {
  const double x = lane_val(1) * 0.5, y = lane_val(2);
  d4 acc[NACC];
  double v[VPER];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < VPER; ++i) v[i] = lane_val(3 + i);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < VPER / NACC; ++j) v[i * (VPER / NACC) + j] = __builtin_fma(v[i * (VPER / NACC) + j], x, y);
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
  for (int i = 0; i < VPER; ++i) s += v[i];
  if (s == 12345.678) out[8] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(w1 - w0); }
}

static double* d_out;
static double h_out[16];

template <typename F>
static float run(F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();  // warm
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
  return best;
}

static void report(const char* name, int nacc, int wps, double flops, float ms, double ninstr_per_wave) {
  const double cyc = h_out[0] / ninstr_per_wave, ghz = h_out[0] / (h_out[1] * 10.0) ;  // 100 MHz ticks -> ns*10
  printf("{\"bench\": \"%s\", \"nacc\": %d, \"waves_per_simd\": %d, \"tflops\": %.2f, \"cycles_per_instr_per_wave\": %.2f, "
         "\"eff_clock_ghz\": %.3f, \"ms\": %.3f}\n", name, nacc, wps, flops / ms / 1e9, cyc, ghz, ms);
  fflush(stdout);
}

int main() {
  CK(hipMalloc(&d_out, sizeof(h_out)));
  CK(hipMemset(d_out, 0, sizeof(h_out)));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, cus, prop.clockRate);
  const int iters = 40000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = cus * wps;
#define MF16(N) { float ms = run([&] { hipLaunchKernelGGL(k_mfma16<N>, dim3(blocks), dim3(256), 0, 0, iters, d_out); }); \
                  report("mfma_f64_16x16x4", N, wps, (double)blocks * 4 * iters * N * 2048.0, ms, (double)iters * N); }
    MF16(2) MF16(4) MF16(8) MF16(16)
#define MF4(N) { float ms = run([&] { hipLaunchKernelGGL(k_mfma4<N>, dim3(blocks), dim3(256), 0, 0, iters, d_out); }); \
                 report("mfma_f64_4x4x4_4b", N, wps, (double)blocks * 4 * iters * N * 512.0, ms, (double)iters * N); }
    MF4(4) MF4(8) MF4(16)
#define VF(N) { float ms = run([&] { hipLaunchKernelGGL(k_valu<N>, dim3(blocks), dim3(256), 0, 0, iters, d_out); }); \
                report("v_fma_f64", N, wps, (double)blocks * 256 * iters * N * 2.0, ms, (double)iters * N); }
    VF(8) VF(16)
  }
  // mixed co-resident waves: 1 MFMA wave + 1 VALU wave per SIMD (512-thread blocks, one per CU), and x2
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int blocks = cus * bpc;
    for (int vit = 0; vit <= 2; ++vit) {
      // viters chosen so both halves run about equally long if the pipes are independent:
      // 8 MFMA x 64 cyc = 512 cyc per iteration; 16 VALU fma x 4 cyc = 64 cyc per iteration
      const int viters = vit == 0 ? 0 : (vit == 1 ? iters * 4 : iters * 8);
      float ms = run([&] { hipLaunchKernelGGL(k_mixed<8>, dim3(blocks), dim3(512), 0, 0, iters, viters, d_out); });
      const double fm = (double)blocks * 4 * iters * 8 * 2048.0, fv = (double)blocks * 256 * (double)viters * 16 * 2.0;
      const double ghz = h_out[0] / (h_out[1] * 10.0);
      printf("{\"bench\": \"mixed_waves\", \"blocks_per_cu\": %d, \"viters_ratio\": %d, \"mfma_tflops_alone_equiv\": %.2f, "
             "\"total_tflops\": %.2f, \"mfma_wave_cycles_per_mfma\": %.2f, \"valu_wave_cycles_per_fma\": %.2f, "
             "\"eff_clock_ghz\": %.3f, \"ms\": %.3f}\n",
             bpc, viters / iters, fm / ms / 1e9, (fm + fv) / ms / 1e9, h_out[0] / ((double)iters * 8),
             viters ? h_out[2] / ((double)viters * 16) : 0.0, ghz, ms);
      fflush(stdout);
    }
  }
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = cus * wps;
#define IN(N, V) { float ms = run([&] { hipLaunchKernelGGL((k_inter<N, V>), dim3(blocks), dim3(256), 0, 0, iters, d_out); }); \
                   const double fl = (double)blocks * 4 * iters * (N * 2048.0 + V * 128.0); \
                   char nm[64]; snprintf(nm, sizeof nm, "interleave_mfma%d_valu%d", N, V); \
                   report(nm, N, wps, fl, ms, (double)iters * N); }
    IN(8, 8) IN(8, 16) IN(8, 32) IN(8, 64)
  }
  return 0;
}
