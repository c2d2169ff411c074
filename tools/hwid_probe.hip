// Which physical compute unit does a workgroup run on?  Census of (XCC_ID, SE_ID, SH_ID, CU_ID) over a launch that
// fills the chip (2 workgroups per CU by LDS), plus the dispatch order -> CU mapping of the first workgroups.
// Evidence for the software CU reservation of the persistent GEMM kernels (gemm.hip).   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin_us) {
  extern __shared__ double lds[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    lds[0] = 1.0;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 100LL * spin_us) __builtin_amdgcn_s_sleep(8);
}

int main() {
  const int n = 1024;
  unsigned* d = nullptr;
  hipMalloc(&d, sizeof(unsigned) * 2 * n);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  std::vector<unsigned> h(2 * n);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(d, 0xff, sizeof(unsigned) * 2 * n);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 80 * 1024, 0, d, 200);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
  }
  std::map<unsigned, int> count;
  printf("first 40 workgroups: wg xcc se sh cu simd\n");
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, simd = (hw >> 4) & 3;
    if (i < 40) printf("  %3d  %u %u %u %2u %u   raw %08x %08x\n", i, xcc, se, sh, cu, simd, hw, h[2 * i + 1]);
    count[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
  }
  printf("distinct (xcc,se,sh,cu): %zu\n", count.size());
  std::map<unsigned, std::vector<unsigned>> per_xcc;
  for (auto& kv : count) per_xcc[kv.first >> 12].push_back(kv.first & 0xfff);
  for (auto& kv : per_xcc) {
    printf("xcc %u: %zu CUs:", kv.first, kv.second.size());
    for (unsigned c : kv.second) printf(" %u.%u.%u(%d)", (c >> 8) & 7, (c >> 4) & 1, c & 0xf, count[(kv.first << 12) | c]);
    printf("\n");
  }
  return 0;
}
