// Which multi-stream patterns survive hipStreamBeginCapture / EndCapture on this ROCm?  (tools/graph_probe2.py: capturing
// the library's factorisation segfaults in hipStreamEndCapture.)   hipcc --offload-arch=gfx950 -o gcp graph_capture_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
__global__ void k(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }

int run_case(int which) {
  double* d; CK(hipMalloc(&d, 1 << 20));
  hipStream_t o, a, b;
  CK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  if (which == 1 || which == 3) CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi));
  else if (which == 4) { uint32_t mask[8] = {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}; CK(hipExtStreamCreateWithCUMask(&a, 8, mask)); }
  else CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t ev[64];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(o, hipStreamCaptureModeGlobal));
  hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, o, d, 16384);
  CK(hipEventRecord(ev[0], o));
  CK(hipStreamWaitEvent(a, ev[0], 0));
  if (which >= 2) CK(hipStreamWaitEvent(b, ev[0], 0));
  int e = 1;
  for (int p = 0; p < (which >= 2 ? 8 : 1); ++p) {
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, a, d, 16384);
    if (which >= 2) {
      CK(hipEventRecord(ev[e], a)); CK(hipStreamWaitEvent(b, ev[e], 0)); ++e;   // a -> b
      hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, b, d + 32768, 16384);
      CK(hipEventRecord(ev[e], b)); CK(hipStreamWaitEvent(a, ev[e], 0)); ++e;   // b -> a (ping-pong like P <-> Bs)
    }
  }
  CK(hipEventRecord(ev[e], a)); CK(hipStreamWaitEvent(o, ev[e], 0)); ++e;
  if (which >= 2) { CK(hipEventRecord(ev[e], b)); CK(hipStreamWaitEvent(o, ev[e], 0)); ++e; }
  printf("  captured, ending...\n"); fflush(stdout);
  CK(hipStreamEndCapture(o, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, o));
  CK(hipStreamSynchronize(o));
  printf("  OK\n"); fflush(stdout);
  return 0;
}
int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 0;
  const char* names[] = {"fork/join, plain side stream", "fork/join, priority side stream", "two side streams ping-pong (plain)",
                         "ping-pong, one priority stream", "ping-pong, one CU-masked stream"};
  printf("case %d: %s\n", which, names[which]); fflush(stdout);
  return run_case(which);
}
