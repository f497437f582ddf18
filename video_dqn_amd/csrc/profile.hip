// Optional in-library launch profiler: when enabled every kernel launch of libvdqn is bracketed by two HIP
// events recorded on the launch stream; vdqn_profile_collect() returns per-kernel totals (launch count, device
// milliseconds, algorithmic FLOPs and bytes).  bench.py uses it for the live roofline numbers; the numbers are
// cross-checked against `rocprofv3 --kernel-trace --stats` (profiles/).
#include <algorithm>
#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace {
struct Rec {
  int tag;
  hipEvent_t e0, e1;
  double flops, bytes;
  hipStream_t st;
};
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
size_t g_pool_next = 0;
std::vector<std::string> g_tags;
std::map<std::string, int> g_tag_ids;
Rec* g_open = nullptr;
}  // namespace

void vdqn_ensure_dyn_smem(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;  // (kernel, device) -> largest size granted so far
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  size_t& have = done[{kernel, dev}];
  if (bytes > have) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    have = bytes;
  }
}

// test hook (not part of include/vdqn.h): pretend the device has n CUs (0 = the real count).  The persistent kernels size their
// grids by it, so a test can run their tile walks — several tiles per workgroup, XCD ranges of unequal length — on small tensors.
static std::atomic<int> g_cus_override{0};
extern "C" void vdqn_debug_set_num_cus(int n) { g_cus_override.store(n > 0 ? n : 0); }

int vdqn_num_cus() {
  if (const int o = g_cus_override.load()) return o;
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  int& n = cus[dev & 63];
  if (n == 0) {
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return n;
}

thread_local double g_prof_alg_flops = -1.0;
thread_local const char* g_prof_suffix = nullptr;  // engine: layer name of the next launch (VDQN_PROFILE_LAYERS=1)

void vdqn_prof_begin(const char* tag, double flops, double bytes, hipStream_t st) {
  if (!g_on) {
    g_prof_alg_flops = -1.0;
    return;
  }
  if (g_prof_alg_flops >= 0.0) flops = g_prof_alg_flops;
  g_prof_alg_flops = -1.0;
  if (g_pool_next == g_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_pool.emplace_back(a, b);
  }
  std::string full(tag);
  if (g_prof_suffix) {
    // keep the layer name when the 47-character entry name cannot hold both
    std::string sfx(g_prof_suffix);
    if (full.size() + 1 + sfx.size() > 47) full = full.substr(0, 47 - 1 - std::min<size_t>(sfx.size(), 46));
    full += "|" + sfx;
    g_prof_suffix = nullptr;
  }
  tag = full.c_str();
  auto it = g_tag_ids.find(tag);
  int id;
  if (it == g_tag_ids.end()) {
    id = (int)g_tags.size();
    g_tags.emplace_back(tag);
    g_tag_ids[tag] = id;
  } else {
    id = it->second;
  }
  g_recs.push_back(Rec{id, g_pool[g_pool_next].first, g_pool[g_pool_next].second, flops, bytes, st});
  ++g_pool_next;
  g_open = &g_recs.back();
  (void)hipEventRecord(g_open->e0, st);
}

void vdqn_prof_end(hipStream_t st) {
  if (!g_on || !g_open) return;
  (void)hipEventRecord(g_open->e1, st);
  g_open = nullptr;
}

extern "C" int vdqn_profile_enable(int on) {
  g_on = on != 0;
  g_recs.clear();
  g_pool_next = 0;
  g_open = nullptr;
  return VDQN_OK;
}

extern "C" int vdqn_profile_collect(vdqn_prof_entry* out, int max_entries) {
  if (!out || max_entries <= 0) return 0;
  std::vector<vdqn_prof_entry> acc(g_tags.size());
  for (size_t i = 0; i < g_tags.size(); ++i) {
    memset(&acc[i], 0, sizeof(vdqn_prof_entry));
    snprintf(acc[i].name, sizeof(acc[i].name), "%s", g_tags[i].c_str());
  }
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    acc[r.tag].launches += 1;
    acc[r.tag].ms += ms;
    acc[r.tag].flops += r.flops;
    acc[r.tag].bytes += r.bytes;
  }
  int n = 0;
  for (auto& a : acc)
    if (a.launches > 0 && n < max_entries) out[n++] = a;
  g_recs.clear();
  g_pool_next = 0;
  return n;
}

// Diagnostic (not part of include/vdqn.h; tools/timeline_live.py): the recorded launches as spans on a common time axis — e0 / e1
// of every launch relative to the first launch's e0, and a small integer per stream in order of first use.  With the side stream
// on this is the update's real timeline (rocprofv3's kernel trace serialises the dispatches): what overlaps, where the device idles.
// Leaves the records in place (vdqn_profile_collect clears them).
struct vdqn_prof_span {
  char name[48];
  float t0_ms, t1_ms;
  int stream;
};
extern "C" int vdqn_debug_profile_timeline(vdqn_prof_span* out, int max_entries) {
  if (!out || max_entries <= 0 || g_recs.empty()) return 0;
  std::vector<hipStream_t> streams;
  int n = 0;
  for (auto& r : g_recs) {
    if (n >= max_entries) break;
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, g_recs[0].e0, r.e0) != hipSuccess || hipEventElapsedTime(&b, g_recs[0].e0, r.e1) != hipSuccess) continue;
    size_t si = std::find(streams.begin(), streams.end(), r.st) - streams.begin();
    if (si == streams.size()) streams.push_back(r.st);
    snprintf(out[n].name, sizeof(out[n].name), "%s", g_tags[r.tag].c_str());
    out[n].t0_ms = a; out[n].t1_ms = b; out[n].stream = (int)si;
    ++n;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------
// Diagnostic (not part of include/vdqn.h; bench.py reports it next to the roofline): the shader clock this device holds under a
// full-rate bf16 MFMA stream on random operands, read in-kernel as delta s_memtime / delta s_memrealtime (x 100 MHz) —
// tools/probes/clock_calib.hip shows that s_memtime counts shader cycles (1.017 ticks per MFMA cycle) and that the
// GRBM_GUI_ACTIVE quotient of short dispatches reads high.  One workgroup of 256 threads per CU, operands in registers,
// ~10 ms of back-to-back v_mfma_f32_32x32x16_bf16 after ~40 ms of the same as heat.  Allocates and frees 3 MB of device memory.
// ---------------------------------------------------------------------------------------------------------
namespace {
typedef __attribute__((ext_vector_type(16))) float clk_f32x16;
__global__ __launch_bounds__(256, 1) void mfma_clock_kernel(const uint32_t* __restrict__ data, unsigned long long* __restrict__ stamps, float* sink,
                                                            int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, reinterpret_cast<const uint4*>(data)[(blockIdx.x * 256 + tid) * 8 + i]);
    b[i] = __builtin_bit_cast(bf16x8, reinterpret_cast<const uint4*>(data)[(blockIdx.x * 256 + tid) * 8 + 4 + i]);
  }
  clk_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < iters; ++k) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * (rep & 1)], b[j + 2 * (rep >> 1)], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  asm volatile("" : "+v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((tid & 63) == 0) {
    unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 2;
    o[0] = t1 - t0;
    o[1] = r1 - r0;
  }
  if (s == 1.2345e33f) sink[blockIdx.x * 256 + tid] = s;
}
}  // namespace

extern "C" int vdqn_debug_mfma_clock(double* ghz_out, double* tflops_out) {
  const int blocks = vdqn_num_cus();
  const size_t words = (size_t)blocks * 256 * 8 * 4;
  uint32_t* d_data = nullptr;
  unsigned long long* d_st = nullptr;
  float* d_sink = nullptr;
  if (hipMalloc(&d_data, words * 4) != hipSuccess || hipMalloc(&d_st, (size_t)blocks * 8 * 8) != hipSuccess || hipMalloc(&d_sink, (size_t)blocks * 256 * 4) != hipSuccess) {
    vdqn_set_error("vdqn_debug_mfma_clock: hipMalloc failed");
    return VDQN_ERR_LAUNCH;
  }
  std::vector<uint32_t> h(words);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < words; ++i) {  // two bf16 in [-1, 1) per word: random sign and mantissa, exponent 0x3f0 - {0..3}
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const uint32_t lo = (uint32_t)(x & 0x807f) | (0x3f00 - (((uint32_t)(x >> 20) & 3) << 7));
    const uint32_t hi = (uint32_t)((x >> 32) & 0x807f) | (0x3f00 - (((uint32_t)(x >> 52) & 3) << 7));
    h[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(d_data, h.data(), words * 4, hipMemcpyHostToDevice);
  const int iters = 40000;  // 40000 x 512 cycles = ~11 ms at 1.8 GHz
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(mfma_clock_kernel, dim3(blocks), dim3(256), 0, nullptr, d_data, d_st, d_sink, iters);
  (void)hipEventRecord(e0, nullptr);
  hipLaunchKernelGGL(mfma_clock_kernel, dim3(blocks), dim3(256), 0, nullptr, d_data, d_st, d_sink, iters);
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> st((size_t)blocks * 8);
  (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int w = 0; w < blocks * 4; ++w)
    if (st[2 * w + 1] > 0) ghz.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  if (ghz_out) *ghz_out = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
  if (tflops_out) *tflops_out = ms > 0.f ? (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 16) / ms / 1e9 : 0.0;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d_data);
  (void)hipFree(d_st);
  (void)hipFree(d_sink);
  return VDQN_OK;
}
