// hipStreamWaitValue32 / hipStreamWriteValue32 against memory written from inside a running kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorName(e), __LINE__); } } while (0)
__global__ void producer(int* flag, long long delay_ticks) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < delay_ticks) {}
  __hip_atomic_store(flag, 7, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  t0 = wall_clock64();
  while (wall_clock64() - t0 < delay_ticks) {}   // keep running after publishing
}
__global__ void consumer(long long* out) { out[0] = wall_clock64(); }
__global__ void poller(const int* flag, long long* out) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 9) __builtin_amdgcn_s_sleep(4);
  out[1] = wall_clock64();
}
int main() {
  for (int kind = 0; kind < 2; ++kind) {
    int* flag = nullptr;
    if (kind == 0) CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    else CK(hipMalloc((void**)&flag, 8));
    CK(hipMemset(flag, 0, 8));
    long long* out; CK(hipMalloc((void**)&out, 64)); CK(hipMemset(out, 0, 64));
    hipStream_t A, B, C; CK(hipStreamCreate(&A)); CK(hipStreamCreate(&B)); CK(hipStreamCreate(&C));
    hipLaunchKernelGGL(poller, dim3(1), dim3(64), 0, C, flag, out);                // waits for the stream write below
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, A, flag, 200000LL);        // 2 ms, publishes 7, runs 2 ms more
    hipError_t w = hipStreamWaitValue32(B, flag, 7, hipStreamWaitValueGte, 0xffffffffu);
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, B, out);
    hipError_t wr = hipStreamWriteValue32(B, flag, 9, 0);
    hipError_t s = hipDeviceSynchronize();
    long long h[8]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
    printf("kind %d (%s): wait %s write %s sync %s consumer_ran %d poller_saw_write %d\n", kind, kind == 0 ? "signal memory" : "hipMalloc",
           hipGetErrorName(w), hipGetErrorName(wr), hipGetErrorName(s), h[0] != 0, h[1] != 0);
  }
  return 0;
}
