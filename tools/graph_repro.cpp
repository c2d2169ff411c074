// Repro harness for the hipGraph path of gpk_potrf: identity-like SPD matrix, call three times on a created stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/gpk.h"
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048, extra = argc > 2 ? atoi(argv[2]) : 8192;
  hipStream_t S; hipStreamCreate(&S);
  std::vector<double> h((size_t)(n + extra) * n, 0.01);
  for (int i = 0; i < n; ++i) h[(size_t)i * n + i] = 10.0 + i * 1e-3;
  double *A, *invd; int* info;
  hipMalloc(&A, h.size() * 8); hipMalloc(&invd, gpk_invd_elems(n, 1) * 8); hipMalloc(&info, 4);
  for (int it = 0; it < 4; ++it) {
    hipMemcpyAsync(A, h.data(), h.size() * 8, hipMemcpyHostToDevice, S);
    fprintf(stderr, "call %d\n", it);
    int rc = gpk_potrf(S, A, n, extra, n, 1, 0, invd, 0, info);
    fprintf(stderr, "  rc %d\n", rc);
    hipError_t e = hipStreamSynchronize(S);
    int hi = -1; hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost);
    double l00; hipMemcpy(&l00, A, 8, hipMemcpyDeviceToHost);
    fprintf(stderr, "  sync %d info %d L00 %.6f\n", (int)e, hi, l00);
  }
  return 0;
}
