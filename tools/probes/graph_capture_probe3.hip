// Third stage: the library's own gpk_potrf / gpk_svgp_elbo_shard under hipStreamBeginCapture, without torch.
//   hipcc --offload-arch=gfx950 -I include -o gcp3 graph_capture_probe3.hip -L gpflow_amd -lgpk -Wl,-rpath,$PWD/gpflow_amd
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpk.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
#define GK(x) do { int r_ = (x); if (r_) { printf("  %s -> gpk error %d\n", #x, r_); fflush(stdout); return 1; } } while (0)
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048, extra = argc > 2 ? atoi(argv[2]) : 0;
  printf("gpk_potrf n=%d extra=%d under capture\n", n, extra); fflush(stdout);
  const int d = 8;
  std::vector<double> hx((size_t)(n + extra) * d);
  srand(1);
  for (auto& v : hx) v = (rand() / (double)RAND_MAX) * 2 - 1;
  double *X, *T, *T0, *invd; int* info;
  const long lda = n;
  CK(hipMalloc(&X, hx.size() * 8)); CK(hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&T, (size_t)(n + extra) * lda * 8)); CK(hipMalloc(&T0, (size_t)(n + extra) * lda * 8));
  CK(hipMalloc(&invd, gpk_invd_elems(n, 1) * 8)); CK(hipMalloc(&info, 64));
  hipStream_t S; CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  const double ls = 2.0;
  GK(gpk_kernel_matrix(S, 0, X, n, d, nullptr, 0, 0, d, &ls, 0, 1.0, 1.0, 0, T0, lda));
  if (extra) GK(gpk_kernel_matrix(S, 0, X + (long)n * d, extra, d, X, n, d, d, &ls, 0, 1.0, 0.0, 0, T0 + (long)n * lda, lda));
  auto step = [&]() -> int {
    CK(hipMemcpyAsync(T, T0, (size_t)(n + extra) * lda * 8, hipMemcpyDeviceToDevice, S));
    GK(gpk_potrf(S, T, n, extra, lda, 1, 0, invd, 1, info));
    return 0;
  };
  for (int i = 0; i < 3; ++i) if (step()) return 1;
  CK(hipStreamSynchronize(S));
  auto timeit = [&](auto&& f, int reps) { CK(hipStreamSynchronize(S)); auto t0 = std::chrono::steady_clock::now(); for (int i = 0; i < reps; ++i) if (f()) return 1; CK(hipStreamSynchronize(S)); printf("  %.4f ms per call\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps); fflush(stdout); return 0; };
  printf(" eager:\n"); for (int r = 0; r < 3; ++r) if (timeit(step, 40)) return 1;
  std::vector<double> ref((size_t)n * lda), got((size_t)n * lda);
  CK(hipMemcpy(ref.data(), T, ref.size() * 8, hipMemcpyDeviceToHost));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeGlobal));
  if (step()) return 1;
  printf("  captured, ending...\n"); fflush(stdout);
  CK(hipStreamEndCapture(S, &g));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("  %zu nodes\n", nn); fflush(stdout);
  {
    // priorities: what the capture recorded, and (argv[3] = 1) the chain's kernels -- 512-thread workgroups: leaf, one-shot
    // solve / strip -- raised to the highest priority before instantiation
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(g, nodes.data(), &nn));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    int nk = 0, nprio = 0, nset = 0;
    for (auto nd : nodes) {
      hipGraphNodeType ty; CK(hipGraphNodeGetType(nd, &ty));
      if (ty != hipGraphNodeTypeKernel) continue;
      ++nk;
      hipKernelNodeAttrValue v{};
      if (hipGraphKernelNodeGetAttribute(nd, hipKernelNodeAttributePriority, &v) == hipSuccess && v.priority != 0) ++nprio;
      hipKernelNodeParams kp{}; CK(hipGraphKernelNodeGetParams(nd, &kp));
      if (argc > 3 && atoi(argv[3]) == 1 && kp.blockDim.x == 512) {
        v.priority = hi;
        if (hipGraphKernelNodeSetAttribute(nd, hipKernelNodeAttributePriority, &v) == hipSuccess) ++nset;
      }
    }
    printf("  %d kernel nodes, %d captured with a non-default priority, %d raised (range %d..%d)\n", nk, nprio, nset, lo, hi);
    fflush(stdout);
  }
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto replay = [&]() -> int { CK(hipGraphLaunch(ge, S)); return 0; };
  if (replay()) return 1;
  CK(hipStreamSynchronize(S));
  CK(hipMemcpy(got.data(), T, got.size() * 8, hipMemcpyDeviceToHost));
  size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += (ref[i] != got[i]);
  printf("  graph result: %zu of %zu entries differ from the eager factor\n", bad, ref.size());
  printf(" graph replay:\n"); for (int r = 0; r < 3; ++r) if (timeit(replay, 40)) return 1;
  printf(" eager again:\n"); if (timeit(step, 40)) return 1;
  printf("  OK\n");
  return 0;
}
