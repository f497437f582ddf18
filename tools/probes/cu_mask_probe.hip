// Probe: what does a CU mask on a HIP stream (hipExtStreamCreateWithCUMask) select on this MI355X?  Launches 4096 one-wave
// workgroups that spin ~20 us each on a masked stream and records where each ran (HW_REG_XCC_ID, and SE / CU id from HW_REG_HW_ID),
// for a few masks: the first 128 bits, every other bit, the bits of the first 32-bit word only.  Prints, per mask, the number of
// distinct (XCC, CU) places used and the workgroups per XCC.
// Build: hipcc -O3 --offload-arch=gfx950 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}  // 20 us at 100 MHz: long enough for the grid to spread over every allowed CU
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) {
    printf("%-28s stream creation failed: %s\n", name, hipGetErrorString(e));
    return;
  }
  const int n = 4096;
  uint32_t* d;
  hipMalloc(&d, 2 * n * sizeof(uint32_t));
  hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, st, d);
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(2 * n);
  hipMemcpy(h.data(), d, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost);
  std::map<uint32_t, int> per_xcc;
  std::set<std::pair<uint32_t, uint32_t>> places;
  for (int i = 0; i < n; ++i) {
    const uint32_t xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    per_xcc[xcc]++;
    places.insert({xcc, (se << 8) | (sh << 4) | cu});
  }
  printf("%-28s places %3zu  per XCC:", name, places.size());
  for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
  printf("\n");
  hipFree(d);
  hipStreamDestroy(st);
}

int main() {
  run("no mask", {});
  run("first 128 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
  run("last 128 bits", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
  run("every other bit (0x5555..)", std::vector<uint32_t>(8, 0x55555555u));
  run("first word only", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 0-7 of every word", std::vector<uint32_t>(8, 0x000000ffu));
  run("one word of 32 bits", {0xffffffffu});
  run("one word, low 16 bits", {0x0000ffffu});
  return 0;
}
