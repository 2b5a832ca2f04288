// Occupancy probe: how many workgroups of the fused top-K kernels fit a CU.
#include <cstdio>
#include "../recommend_amd/csrc/topk_kernel.h"
template <int KPAD, int DC, bool F> void report(const char *name) {
  int n = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, score_topk_kernel<KPAD, DC, F>, 256, 0);
  hipFuncAttributes a{};
  hipFuncGetAttributes(&a, reinterpret_cast<const void *>(score_topk_kernel<KPAD, DC, F>));
  printf("%s: blocks/CU %d (err %d) regs %d lds %zu scratch %zu\n", name, n, (int)e, a.numRegs, a.sharedSizeBytes, a.localSizeBytes);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d, LDS/CU %zu, LDS/block %zu, regs/block %d\n", p.multiProcessorCount, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerBlock);
  report<28, 1, true>("filter 28 D64"); report<24, 1, true>("filter 24 D64"); report<28, 2, true>("filter 28 D128");
  report<20, 1, false>("exact 20 D64"); report<20, 2, false>("exact 20 D128");
  return 0;
}
