// Diagnostic: which CU does block b of a 2-workgroups-per-CU launch land on?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(512, 2) void k(unsigned* out) {
  extern __shared__ char lds[];
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  long long t = __builtin_amdgcn_s_memtime();
  lds[threadIdx.x] = 1;
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = hw; out[3 * blockIdx.x + 1] = xcc; out[3 * blockIdx.x + 2] = (unsigned)(t >> 6); }
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);
}
int main() {
  int nb = 1024; unsigned* d; hipMalloc(&d, nb * 12);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 75776);
  hipLaunchKernelGGL(k, dim3(nb), dim3(512), 75776, 0, d);
  std::vector<unsigned> h(nb * 3); hipMemcpy(h.data(), d, nb * 12, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu2blocks;
  for (int b = 0; b < nb; ++b) {
    unsigned hw = h[3 * b], xcc = h[3 * b + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cu2blocks[key].push_back(b);
    if (b < 24) printf("block %d: xcc %u se %u sh %u cu %u  t %u\n", b, xcc, se, sh, cu, h[3 * b + 2]);
  }
  printf("distinct CUs %zu\n", cu2blocks.size());
  int shown = 0;
  for (auto& kv : cu2blocks) { if (shown++ < 6) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
  return 0;
}
