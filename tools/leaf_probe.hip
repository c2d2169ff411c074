// Stand-alone probe of the leaf kernels (round 6): correctness against a host Cholesky / inverse for full and ragged blocks,
// the non-positive-pivot report, in-kernel phase timers and the per-launch time of back-to-back launches, for
//   v1  leaf_device.h   (rounds 1 - 5)          v2  leaf2_device.h  (round 6)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on tools/leaf_probe.hip -o /tmp/leaf_probe && /tmp/leaf_probe
#include "../gpflow_amd/csrc/leaf2_device.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace gpk_leaf;

template <int V>
__global__ __launch_bounds__(V == 1 ? NT : gpk_leaf2::NT2) void k_leaf(double* A, long lda, int nb, double* inv, int* info, long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) double S[];
  if constexpr (V == 1) leaf_body<false>(S, A, lda, nb, inv, info, 0, dbg);
  else if constexpr (V == 2) gpk_leaf2::leaf2_body<false>(S, A, lda, nb, inv, info, 0, dbg);
  else gpk_leaf2::leaf2_body<true>(S, A, lda, nb, inv, info, 0, dbg);
}

__global__ void k_hwid(int* out) {
  // HW_ID register (id 4): SIMD_ID bits [5:4], CU_ID [11:8], SE_ID [15:13] on gfx9
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (int)v;
}

static void host_chol(const std::vector<double>& A, int n, std::vector<double>& L, std::vector<double>& X) {
  L.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    long double s = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= (long double)L[(size_t)j * n + k] * L[(size_t)j * n + k];
    const long double ljj = sqrtl(s);
    L[(size_t)j * n + j] = (double)ljj;
    for (int i = j + 1; i < n; ++i) {
      long double t = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= (long double)L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = (double)(t / ljj);
    }
  }
  X.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    X[(size_t)j * n + j] = 1.0 / L[(size_t)j * n + j];
    for (int i = j + 1; i < n; ++i) {
      long double t = 0;
      for (int k = j; k < i; ++k) t += (long double)L[(size_t)i * n + k] * X[(size_t)k * n + j];
      X[(size_t)i * n + j] = (double)(-t / L[(size_t)i * n + i]);
    }
  }
}

template <int V>
static void launch(double* dA, long lda, int nb, double* dinv, int* dinfo, long long* ddbg) {
  hipLaunchKernelGGL((k_leaf<V>), dim3(1), dim3(V == 1 ? NT : gpk_leaf2::NT2), gpk_leaf2::LEAF2_LDS, 0, dA, lda, nb, dinv, dinfo, ddbg);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_leaf<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gpk_leaf2::LEAF2_LDS));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_leaf<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gpk_leaf2::LEAF2_LDS));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_leaf<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gpk_leaf2::LEAF2_LDS));
  const int lda = 136;
  double *dA, *dinv; int* dinfo; long long* ddbg; int* dhw;
  CK(hipMalloc(&dA, sizeof(double) * 128 * lda)); CK(hipMalloc(&dinv, sizeof(double) * 128 * 128));
  CK(hipMalloc(&dinfo, sizeof(int))); CK(hipMalloc(&ddbg, sizeof(long long) * 320)); CK(hipMemset(ddbg, 0, sizeof(long long) * 320)); CK(hipMalloc(&dhw, sizeof(int) * 16));
  hipLaunchKernelGGL(k_hwid, dim3(1), dim3(768), 0, 0, dhw);
  int hw[12]; CK(hipMemcpy(hw, dhw, sizeof(hw), hipMemcpyDeviceToHost));
  printf("# HW_ID of waves 0..11 of a 768-thread block: SIMD ids");
  for (int w = 0; w < 12; ++w) printf(" %d", (hw[w] >> 4) & 3);
  printf("\n");
  int fails = 0;
  const int nbs[] = {128, 127, 113, 112, 100, 64, 33, 17, 16, 5, 1};
  for (int nb : nbs) {
    std::vector<double> A((size_t)nb * nb), L, X;
    srand(1234 + nb);
    std::vector<double> B((size_t)nb * nb);
    for (auto& v : B) v = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j < nb; ++j) {
        double s = (i == j) ? 0.05 * nb : 0.0;
        for (int k = 0; k < nb; ++k) s += B[(size_t)i * nb + k] * B[(size_t)j * nb + k];
        A[(size_t)i * nb + j] = s;
      }
    host_chol(A, nb, L, X);
    for (int v = 1; v <= 2; ++v) {
      std::vector<double> hA((size_t)128 * lda, -7.0), hinv((size_t)128 * 128, -9.0);   // sentinels: nothing outside the block may change
      for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) hA[(size_t)i * lda + j] = A[(size_t)i * nb + j];
      CK(hipMemcpy(dA, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
      CK(hipMemcpy(dinv, hinv.data(), sizeof(double) * hinv.size(), hipMemcpyHostToDevice));
      CK(hipMemset(dinfo, 0, sizeof(int)));
      if (v == 1) launch<1>(dA, lda, nb, dinv, dinfo, nullptr); else launch<2>(dA, lda, nb, dinv, dinfo, nullptr);
      CK(hipDeviceSynchronize());
      std::vector<double> oA(hA.size()), oinv(hinv.size());
      int info = -1;
      CK(hipMemcpy(oA.data(), dA, sizeof(double) * oA.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(oinv.data(), dinv, sizeof(double) * oinv.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost));
      double eL = 0, eX = 0, mL = 0, mX = 0; int touched = 0;
      for (int i = 0; i < 128; ++i)
        for (int j = 0; j < lda; ++j) {
          const double got = oA[(size_t)i * lda + j];
          if (i < nb && j <= i) { eL = fmax(eL, fabs(got - L[(size_t)i * nb + j])); mL = fmax(mL, fabs(L[(size_t)i * nb + j])); }
          else if (got != hA[(size_t)i * lda + j]) ++touched;   // strict upper part / padding must be left alone
        }
      for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
          const double want = (i < nb && j < nb) ? X[(size_t)i * nb + j] : (i == j ? 1.0 : 0.0);
          eX = fmax(eX, fabs(oinv[(size_t)i * 128 + j] - want)); mX = fmax(mX, fabs(want));
        }
      const bool ok = eL / mL < 1e-13 && eX / mX < 1e-12 && touched == 0 && info == 0;
      printf("v%d nb=%3d  |L-chol|/|L| = %.2e  |inv-L^-1|/|inv| = %.2e  touched-outside = %d  info = %d  %s\n", v, nb, eL / mL, eX / mX,
             touched, info, ok ? "ok" : "FAIL");
      if (!ok) ++fails;
    }
  }
  // non-positive pivot: A[77][77] made very negative -> info = 78; NaN input -> info = its column + 1
  for (int v = 1; v <= 2; ++v)
    for (int bad : {0, 3, 77, 127}) {
      std::vector<double> hA((size_t)128 * lda, 0.0);
      for (int i = 0; i < 128; ++i) hA[(size_t)i * lda + i] = 2.0 + 0.01 * i;
      for (int i = 1; i < 128; ++i) hA[(size_t)i * lda + i - 1] = 0.5;
      hA[(size_t)bad * lda + bad] = -3.0;
      CK(hipMemcpy(dA, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
      CK(hipMemset(dinfo, 0, sizeof(int)));
      if (v == 1) launch<1>(dA, lda, 128, dinv, dinfo, nullptr); else launch<2>(dA, lda, 128, dinv, dinfo, nullptr);
      int info = -1;
      CK(hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost));
      printf("v%d bad pivot at %3d -> info %d %s\n", v, bad, info, info == bad + 1 ? "ok" : "FAIL");
      if (info != bad + 1) ++fails;
    }
  // timing: full block
  {
    const int nb = 128;
    std::vector<double> hA((size_t)128 * lda, 0.0);
    srand(7);
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j <= i; ++j) hA[(size_t)i * lda + j] = (i == j) ? 40.0 + i : ((double)rand() / RAND_MAX - 0.5);
    double* dA0;
    CK(hipMalloc(&dA0, sizeof(double) * hA.size()));
    CK(hipMemcpy(dA0, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int v = 1; v <= 2; ++v) {
      long long dbg[8];
      for (int r = 0; r < 3; ++r) {
        CK(hipMemcpy(dA, dA0, sizeof(double) * hA.size(), hipMemcpyDeviceToDevice));
        if (v == 1) launch<1>(dA, lda, nb, dinv, dinfo, ddbg); else launch<2>(dA, lda, nb, dinv, dinfo, ddbg);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(dbg, ddbg, sizeof(dbg), hipMemcpyDeviceToHost));
        printf("v%d phases (us, 100 MHz clock): %s %.2f  factor %.2f  invert %.2f  %s %.2f  total %.2f\n", v, v == 1 ? "load" : "-",
               dbg[0] / 100.0, dbg[1] / 100.0, dbg[2] / 100.0, v == 1 ? "store" : "tail", dbg[3] / 100.0, dbg[4] / 100.0);
      }
      // back-to-back launches on one stream (the factor of a factor is meaningless; the time is not data dependent as long as
      // the pivots stay positive: refresh the block with a device copy every launch, timed separately and subtracted)
      float ms_copy = 0, ms_both = 0;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(dA, dA0, sizeof(double) * hA.size(), hipMemcpyDeviceToDevice, 0));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_copy, e0, e1));
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) {
        CK(hipMemcpyAsync(dA, dA0, sizeof(double) * hA.size(), hipMemcpyDeviceToDevice, 0));
        if (v == 1) launch<1>(dA, lda, nb, dinv, dinfo, nullptr); else launch<2>(dA, lda, nb, dinv, dinfo, nullptr);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_both, e0, e1));
      printf("v%d back-to-back: %.2f us per launch (copy + leaf %.2f, copy alone %.2f)\n", v, (ms_both - ms_copy) * 1e3 / reps,
             ms_both * 1e3 / reps, ms_copy * 1e3 / reps);
    }
  }
  // stamps of the round-6 leaf (shader cycles)
  {
    const int nb = 128;
    std::vector<double> hA((size_t)128 * lda, 0.0);
    srand(7);
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j <= i; ++j) hA[(size_t)i * lda + j] = (i == j) ? 40.0 + i : ((double)rand() / RAND_MAX - 0.5);
    for (int r = 0; r < 2; ++r) {
      CK(hipMemcpy(dA, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
      launch<3>(dA, lda, nb, dinv, dinfo, ddbg);
      CK(hipDeviceSynchronize());
    }
    long long st[320];
    CK(hipMemcpy(st, ddbg, sizeof(st), hipMemcpyDeviceToHost));
    const long long z = st[16];
    printf("# pivot wave of tile k, shader cycles: panel 0 | 1 | 2 | 3 + store   (start .. end since the first tile's start)\n");
    for (int k = 0; k < 8; ++k) {
      const long long* q = st + 16 + 8 * k;
      printf("tile %d: %6lld | %6lld | %6lld | %6lld   (%lld .. %lld)\n", k, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[0] - z, q[4] - z);
    }
    printf("# trailing wave of tile k: start at | waits | L(k+1,k-1) | look-ahead pair | follow panels -> s2 final at\n");
    for (int k = 0; k < 8; ++k) {
      const long long* q = st + 80 + 8 * k;
      printf("tile %d: %6lld | %6lld | %6lld | %6lld | %6lld -> %lld\n", k, q[0] - z, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[4] - z);
    }
    printf("# kernel entered at %lld; helper 0: loads issued at %lld, rows in LDS at %lld, all helpers loaded at %lld\n", st[11] - z, st[8] - z, st[9] - z, st[10] - z);
    printf("# helpers 0..9 past the first barrier at:");
    for (int q = 0; q < 10; ++q) printf(" %lld", st[288 + q] - z);
    printf("\n# helpers 0..9 rows in LDS at:");
    for (int q = 0; q < 10; ++q) printf(" %lld", st[272 + q] - z);
    printf("\n");
    for (int w = 0; w < 2; ++w) {
      printf(w == 0 ? "# owner of row 7 of the INVERSE, window t: entered at | - | wait A_t | - | - | duties | (f) inverse sums (done at)\n"
                    : "# owner of row 7 of L, window t: entered at | (a) early terms | wait A_t | (d) L(7,t) | (e) look-ahead tiles | duties | - (done at)\n");
      for (int k = 0; k < 8; ++k) {
        const long long* q = st + 144 + 64 * w + 8 * k;
        printf("window %d: %6lld | %6lld | %6lld | %6lld | %6lld | %6lld | %6lld (%lld)\n", k, q[0] - z, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3],
               q[5] - q[4], q[6] - q[5], q[6] - z);
      }
    }
  }
  printf(fails ? "FAILURES: %d\n" : "all ok (%d)\n", fails);
  return fails ? 1 : 0;
}
