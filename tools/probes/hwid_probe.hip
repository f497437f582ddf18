// Which hardware slots do the two workgroups that share a CU get?  512 workgroups of 256 threads with 80 KiB of LDS each (two per
// CU, as the persistent kernels of libvdqn), every wave records HW_ID and XCC_ID; the host prints, per CU, the blocks it held and
// the wave slots of their waves.   hipcc --offload-arch=gfx950 -O2 -o hwid_probe hwid_probe.hip && ./hwid_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);  // stay resident so that all 512 workgroups coexist
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  if (threadIdx.x == 999) smem[0] = 1;
}
int main() {
  const int grid = 512;
  unsigned* d;
  hipMalloc(&d, grid * 4 * 2 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 80 * 1024, 0, d, 200);
  hipDeviceSynchronize();
  std::vector<unsigned> h(grid * 4 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> per_cu;  // (xcc, se, sh, cu) -> blocks
  int slot_hist[16] = {0};
  int mixed = 0;
  for (int b = 0; b < grid; ++b) {
    unsigned par = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2];
      slot_hist[hw & 15]++;
      par |= 1u << (hw & 1);
    }
    if (par == 3) ++mixed;
    const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
    per_cu[(xcc << 16) | (hw & 0xff00)].push_back(b);
  }
  printf("wave-slot histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", slot_hist[i]); printf("\n");
  printf("workgroups whose four waves have mixed slot parity: %d of %d\n", mixed, grid);
  int same = 0, diff = 0, n_cu = 0, other = 0;
  for (auto& kv : per_cu) {
    ++n_cu;
    if (kv.second.size() != 2) { ++other; continue; }
    const unsigned a = h[kv.second[0] * 8] & 1, b2 = h[kv.second[1] * 8] & 1;
    (a == b2 ? same : diff)++;
  }
  printf("CUs seen %d; with two workgroups: parity differs on %d, equal on %d; other counts %d\n", n_cu, diff, same, other);
  int shown = 0;
  for (auto& kv : per_cu) {
    if (shown++ >= 6) break;
    printf("cu key %06x:", kv.first);
    for (int b : kv.second) {
      printf("  block %d slots", b);
      for (int w = 0; w < 4; ++w) printf(" %u/simd%u", h[(b * 4 + w) * 2] & 15, (h[(b * 4 + w) * 2] >> 4) & 3);
    }
    printf("\n");
  }
  return 0;
}
