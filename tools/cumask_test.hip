// Does hipExtStreamCreateWithCUMask partition the 256 CUs of an MI355X?  Launch a census kernel on a masked
// stream and print which (xcc, se, sh, cu) tuples ran workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void census(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(hipStream_t s, const char* name, int blocks) {
  unsigned* d; CK(hipMalloc(&d, blocks * 8));
  hipLaunchKernelGGL(census, dim3(blocks), dim3(256), 0, s, d, 2000);
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(2 * blocks);
  CK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
  std::set<unsigned> cus; std::set<unsigned> xccs;
  int per_xcc[16] = {0};
  for (int i = 0; i < blocks; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    if (cus.insert(key).second) per_xcc[xcc]++;
    xccs.insert(xcc);
  }
  printf("%s: %d blocks ran on %zu distinct CUs over %zu XCCs; per-XCC CU counts:", name, blocks, cus.size(), xccs.size());
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  CK(hipFree(d));
}

int main() {
  hipStream_t s0; CK(hipStreamCreate(&s0));
  run(s0, "unmasked", 4096);
  for (int variant = 0; variant < 4; ++variant) {
    uint32_t mask[8];
    for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
    const char* nm = "";
    if (variant == 0) { mask[0] = 0xffffff00u; nm = "mask: bits 0..7 cleared"; }
    if (variant == 1) { mask[7] = 0x00ffffffu; nm = "mask: bits 248..255 cleared"; }
    if (variant == 2) { for (int i = 0; i < 8; ++i) mask[i] = 0; mask[0] = 0xffu; nm = "mask: only bits 0..7 set"; }
    if (variant == 3) { for (int i = 0; i < 8; ++i) mask[i] = 0; mask[0] = 0xffffu; nm = "mask: only bits 0..15 set"; }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", nm, hipGetErrorString(e)); continue; }
    run(s, nm, 4096);
    CK(hipStreamDestroy(s));
  }
  return 0;
}
