// Second stage of the hipGraph capture probe: the factorisation's own stream/event pattern (chain P, rest-updates Bs,
// extra rows X; 16 panels), optionally with a 150-KB dynamic-LDS kernel, a memset node, big by-value kernel arguments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
struct Big { double v[40]; const double* p; long n; };
__global__ void k(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
__global__ void kbig(Big b, double* p) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < b.n) p[i] += b.v[i % 40]; }
__global__ void klds(double* p, int n) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = p[threadIdx.x]; __syncthreads();
  p[threadIdx.x] = sm[(threadIdx.x + 1) % blockDim.x] + 1.0;
}
int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 5;
  printf("case %d\n", which); fflush(stdout);
  double* d; CK(hipMalloc(&d, 1 << 22));
  int* info; CK(hipMalloc(&info, 64));
  hipStream_t S, P, X, Bs;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&P, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&X, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&Bs, hipStreamNonBlocking));
  if (which == 6) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(klds), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const int np = 16;
  hipEvent_t ev[2 * np + 8];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t *evF = ev, *evR = ev + np, evFork = ev[2 * np], evJP = ev[2 * np + 1], evJB = ev[2 * np + 2], evJX = ev[2 * np + 3];
  Big big{}; big.n = 1000; big.p = d;
  for (int rep = 0; rep < 2; ++rep) {   // rep 0 eager (like the library's warm-up calls), rep 1 captured
    hipGraph_t g; hipGraphExec_t ge;
    if (rep) CK(hipStreamBeginCapture(S, which == 8 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeGlobal));
    if (which == 7) CK(hipMemsetAsync(info, 0, 16, S));
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, S, d, 16384);
    CK(hipEventRecord(evFork, S));
    CK(hipStreamWaitEvent(P, evFork, 0)); CK(hipStreamWaitEvent(Bs, evFork, 0)); CK(hipStreamWaitEvent(X, evFork, 0));
    int last_rest = -1;
    for (int p = 0; p < np; ++p) {
      if (which == 6) hipLaunchKernelGGL(klds, dim3(1), dim3(512), 150 * 1024, P, d, 512);
      else hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, P, d, 256);
      if (which == 9) hipLaunchKernelGGL(kbig, dim3(4), dim3(256), 0, P, big, d + 4096);
      else hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, P, d + 4096, 1024);
      CK(hipEventRecord(evF[p], P));
      if (p + 1 < np) {
        if (last_rest >= 0) CK(hipStreamWaitEvent(P, evR[last_rest], 0));
        hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, P, d + 8192, 1024);
      }
      if (p + 2 < np) {
        CK(hipStreamWaitEvent(Bs, evF[p], 0));
        hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, Bs, d + 65536, 16384);
        CK(hipEventRecord(evR[p], Bs));
        last_rest = p;
      }
      if ((p & 3) == 3) {
        CK(hipStreamWaitEvent(X, evF[p], 0));
        for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, X, d + 131072, 65536);
      }
    }
    CK(hipEventRecord(evJP, P)); CK(hipStreamWaitEvent(S, evJP, 0));
    CK(hipEventRecord(evJB, Bs)); CK(hipStreamWaitEvent(S, evJB, 0));
    CK(hipEventRecord(evJX, X)); CK(hipStreamWaitEvent(S, evJX, 0));
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, S, d, 16384);
    if (rep) {
      printf("  captured, ending...\n"); fflush(stdout);
      CK(hipStreamEndCapture(S, &g));
      size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("  %zu nodes\n", nn); fflush(stdout);
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, S));
    }
    CK(hipStreamSynchronize(S));
  }
  printf("  OK\n"); fflush(stdout);
  return 0;
}
