// Which stream kinds can join a HIP stream capture through an event fork?  (priority / non-blocking / CU-masked)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
static const char* es(hipError_t e) { return hipGetErrorName(e); }
int main() {
  int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  hipStream_t S; hipStreamCreate(&S);
  for (int kind = 0; kind < 4; ++kind) {
    hipStream_t T = nullptr; hipError_t ce = hipSuccess;
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (kind == 0) ce = hipStreamCreateWithFlags(&T, hipStreamNonBlocking);
    if (kind == 1) ce = hipStreamCreateWithPriority(&T, hipStreamNonBlocking, hi);
    if (kind == 2) ce = hipStreamCreateWithPriority(&T, hipStreamDefault, hi);
    if (kind == 3) { uint32_t m[8]; for (int i = 0; i < 8; ++i) m[i] = 0xffffffffu; m[0] = 0xffff0000u; ce = hipExtStreamCreateWithCUMask(&T, 8, m); }
    hipEvent_t e0, e1; hipEventCreateWithFlags(&e0, hipEventDisableTiming); hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    hipGraph_t g = nullptr; hipGraphExec_t x = nullptr;
    hipError_t a = hipStreamBeginCapture(S, hipStreamCaptureModeRelaxed);
    hipError_t b = hipEventRecord(e0, S);
    hipError_t c = hipStreamWaitEvent(T, e0, 0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, T, d);
    hipError_t l = hipGetLastError();
    hipError_t r = hipEventRecord(e1, T);
    hipError_t w = hipStreamWaitEvent(S, e1, 0);
    hipError_t en = hipStreamEndCapture(S, &g);
    hipError_t in = g ? hipGraphInstantiate(&x, g, nullptr, nullptr, 0) : hipErrorUnknown;
    hipError_t la = x ? hipGraphLaunch(x, S) : hipErrorUnknown;
    hipStreamSynchronize(S);
    printf("kind %d: create %s begin %s rec %s wait %s launch %s rec2 %s wait2 %s end %s inst %s graphlaunch %s\n", kind, es(ce), es(a), es(b), es(c),
           es(l), es(r), es(w), es(en), es(in), es(la));
    (void)hipGetLastError();
  }
  int h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("count %d\n", h);
  return 0;
}
