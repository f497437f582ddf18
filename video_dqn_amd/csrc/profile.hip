// Optional in-library launch profiler: when enabled every kernel launch of libvdqn is bracketed by two HIP
// events recorded on the launch stream; vdqn_profile_collect() returns per-kernel totals (launch count, device
// milliseconds, algorithmic FLOPs and bytes).  bench.py uses it for the live roofline numbers; the numbers are
// cross-checked against `rocprofv3 --kernel-trace --stats` (profiles/).
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace {
struct Rec {
  int tag;
  hipEvent_t e0, e1;
  double flops, bytes;
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

int vdqn_num_cus() {
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
  g_recs.push_back(Rec{id, g_pool[g_pool_next].first, g_pool[g_pool_next].second, flops, bytes});
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
