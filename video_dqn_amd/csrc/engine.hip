// The network engine: HabitatDQNMultiAction (ResNet-18 trunk + extra_capacity head) as a static layer
// table, BatchNorm-eval folding / weight packing, forward, and the staged backward of one TD update.
//
// Follows: archs/HabitatDQNMultiAction.py:9-54 (wiring, set_train: trunk BatchNorm in eval mode),
// torchvision 0.4.2 resnet18 topology (third-party; restated in oracle/ref_cpu.py),
// train_q_network.py:126-181 (process_batch) and :222-227 (zero_grad / backward / step order).
//
// BatchNorm-eval is folded into the packed weights:  y = conv(x, W * s) + (beta - mean * s),  s = gamma * rstd.
// Its parameter gradients need no saved conv output:  with dW' = dL/d(W*s) (what the wgrad kernel produces)
//   dL/dW = dW' * s,   dL/dbeta = sum(gy),   dL/dgamma = rstd * ( <dW'[co,:], W[co,:]> - mean * dL/dbeta ).
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// error text (thread local)
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vdqn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vdqn_last_error(void) { return g_err; }
extern "C" int vdqn_abi_version(void) { return 14; }
extern "C" int32_t vdqn_abi_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(vdqn_conv_args);
    case 1: return (int32_t)sizeof(vdqn_wgrad_args);
    case 2: return (int32_t)sizeof(vdqn_td_args);
    case 3: return (int32_t)sizeof(vdqn_net_config);
    case 4: return (int32_t)sizeof(vdqn_param_info);
    case 5: return (int32_t)sizeof(vdqn_prof_entry);
    case 6: return (int32_t)sizeof(vdqn_step_args);
    default: return -1;
  }
}

namespace {

constexpr int kMaxLayers = 24;
constexpr float kBnEps = 1e-5f;
constexpr float kBnMomentum = 0.1f;

enum LayerKind { K_CONV = 0, K_CONV1_S2D = 1, K_LINEAR = 2, K_LINEAR_PERM = 3 };

struct Layer {
  std::string name, bn_name;
  int kind;
  int co, ci, r, s, stride, pad;  // master weight dims [co][ci][r][s] and conv geometry
  int has_bn, has_bias;
  int k_ci, k_r, k_s, pix_stride;  // kernel view
  int co_pad;
  int hi, wi, ho, wo;
  int per_sample;
  int stage, has_dgrad;
  int64_t w_off, g_off, b_off, mean_off, var_off;
  int64_t wf_off, wd_off, bias_off, scale_off;  // bytes in packed
  int64_t dw_off, db_off;                        // bytes in bwd workspace
  int kf() const { return k_r * k_s * k_ci; }
  int kd() const { return k_r * k_s * co_pad; }
};

struct FoldDesc {
  int64_t w_off, g_off, b_off, mean_off, var_off;
  int64_t wf_off, wd_off, bias_off, scale_off, dw_off, db_off;
  int co, ci, r, s, kind, co_pad, kf, k_ci, k_s, cd_rows, kd, has_bn, has_bias;
  int tiled;       // packed by fold_tile_kernel (plain convs with ci % 64 == 0): 64 x 64-channel tiles through LDS
  int tile_begin;  // first tile of this layer in fold_tile_kernel's grid
};
struct FoldTable {
  int n, n_tiles;
  FoldDesc d[kMaxLayers];
};
// where unfold finds dL/dbias of layer i: tiles > 0 -> sum of dgrad-epilogue partials
//   sum_t sum_g part[t * ld + g * gstride + co]   (g < groups), else the colsum kernel's db[co]
struct PartTable {
  int64_t off[kMaxLayers];
  int tiles[kMaxLayers], ld[kMaxLayers], groups[kMaxLayers], gstride[kMaxLayers];
};

struct ActLayout {
  int64_t t_in, c1, pool, idx;
  int64_t h[8], o[8], ds[8];
  int64_t f8, l0, l1, q, qf;
  // ARCHITECTURE='basic' only: pooled features, raw (pre-BatchNorm) conv outputs, per-layer BatchNorm work areas
  int64_t avg, r_c1, r_h[8], r_o[8], r_ds[8], bnw[kMaxLayers], bnw_begin, bnw_bytes, bn_sync;
  int64_t bn_det = -1, bn_det_bytes = 0;  // deterministic mode ('basic'): per-block partial sums of the train-mode BatchNorm kernels
  int64_t total;
};
struct BwdLayout {
  int64_t zero_begin, zero_bytes;  // region cleared every step: dW', dbias', loss scratch
  int64_t dq, g_l1, g_l0, g_f8, g_o[8], g_h[8], dsg[8], g_pool, g_c1;
  int64_t p_l1, p_l0, p_f8, p_o[8], p_h[8], p_pool;  // per-128-row-tile column sums written by the dgrad epilogues
  int64_t g_avg, g_or[8], g_dsr[8];  // 'basic' only: gradient of the pooled features / of the raw conv2, downsample outputs
  int64_t det_ws, det_ws_bytes;      // deterministic mode: the weight-gradient kernels' partial copies (one layer at a time)
  int64_t total;
};

inline int64_t align_up(int64_t v, int64_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace

struct vdqn_net {
  vdqn_net_config cfg;
  int esz;  // bytes per activation element
  std::vector<Layer> layers;
  std::vector<vdqn_param_info> params;
  int64_t trainable_numel, params_numel, bnstats_numel, packed_bytes;
  int64_t stage_begin[3], stage_end[3];
  int layer_stage_first[3], layer_stage_count[3];
  int64_t dw_bytes;  // total f32 dW' + dbias' bytes
  FoldTable fold;
  // layer indices
  int l_conv1, l_f8, l_top0, l_top2, l_top4;  // 'basic': l_top4 is the single `top` Linear, the other head layers are -1
  bool basic() const { return cfg.extra_capacity == 0; }
  int l_b_conv1[8], l_b_conv2[8], l_b_ds[8];
  // A second HIP stream for work that is independent of the main dependency chain: the weight gradients (they only
  // need gy, the data-gradient chain does not wait for them) and the target-network forward.  Blocks of the side
  // kernels fill the tail rounds of the main kernels (784..3136-block grids on 512 resident blocks).
  BnSync bn_sync = {nullptr, nullptr, nullptr, 1};  // SyncBN hook ('basic' under data parallelism)
  int wgrad_rr = 0;                                  // VDQN_WGRAD_STREAMS=2: which side stream took the last weight gradient
  int overlap = 1;
  int bwd_samples = 0;  // batch of the update in flight (set by vdqn_net_td_forward; sizes the bwd workspace layout)
  hipStream_t side = nullptr;
  hipStream_t side2 = nullptr;  // the second half of the online forward pass
  std::vector<hipEvent_t> events;
  size_t ev_next = 0;
};

namespace {

bool side_ready(vdqn_net* net) {
  if (!net->overlap) return false;
  if (!net->side) {
    // The side streams (target forward, weight gradients, unfold) run BELOW the caller's stream in the hardware queues' priority order:
    // the caller's stream carries the critical chain (online forward, data gradients), the side streams are what fills the chip around
    // it.  Measured on alternating runs of two boxes (profiles/r04i_*, r04j_*): low 5.766-5.781 against 5.832-5.842 ms per update on one
    // (+1.0 %, three rounds of three), 5.843-5.856 against 5.849-5.863 on the other (+0.1 %); HIGH priority for them: 5.911-5.916 against
    // 5.837-5.844 (-1.2 %).  VDQN_SIDE_PRIORITY=normal|high|low overrides.
    int prio = 0;
    {
      const char* e = getenv("VDQN_SIDE_PRIORITY");
      int lo = 0, hi = 0;  // (numerically lower = higher priority)
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) prio = (!e || e[0] == 'l') ? lo : (e[0] == 'h' ? hi : 0);
    }
    if (hipStreamCreateWithPriority(&net->side, hipStreamNonBlocking, prio) != hipSuccess ||
        hipStreamCreateWithPriority(&net->side2, hipStreamNonBlocking, prio) != hipSuccess) {
      net->overlap = 0;
      net->side = nullptr;
      net->side2 = nullptr;
      return false;
    }
    net->events.resize(64);
    for (auto& e : net->events)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        net->overlap = 0;
        return false;
      }
  }
  return true;
}
hipEvent_t next_event(vdqn_net* net) {
  hipEvent_t e = net->events[net->ev_next];
  net->ev_next = (net->ev_next + 1) % net->events.size();
  return e;
}
// the side stream waits for everything queued on `main` so far; returns the stream to launch the side work on
hipStream_t fork_side(vdqn_net* net, hipStream_t main) {
  if (!side_ready(net)) return main;
  hipEvent_t e = next_event(net);
  (void)hipEventRecord(e, main);
  (void)hipStreamWaitEvent(net->side, e, 0);
  return net->side;
}
// `main` waits for everything queued on the side stream so far
void join_side(vdqn_net* net, hipStream_t main) {
  if (!net->overlap || !net->side) return;
  hipEvent_t e = next_event(net);
  (void)hipEventRecord(e, net->side);
  (void)hipStreamWaitEvent(main, e, 0);
}
// the same pair for the second side stream
hipStream_t fork_side2(vdqn_net* net, hipStream_t main) {
  if (!side_ready(net)) return main;
  hipEvent_t e = next_event(net);
  (void)hipEventRecord(e, main);
  (void)hipStreamWaitEvent(net->side2, e, 0);
  return net->side2;
}
void join_side2(vdqn_net* net, hipStream_t main) {
  if (!net->overlap || !net->side2) return;
  hipEvent_t e = next_event(net);
  (void)hipEventRecord(e, net->side2);
  (void)hipStreamWaitEvent(main, e, 0);
}

// Stream of the next weight-gradient launch.  The launches alternate between the two side streams, so that one kernel's tail (a
// single round of split-K blocks that all end in f32 atomics) runs beside the next kernel's start; not in the deterministic /
// two-stage modes, whose partial copies share one workspace.  The stage's unfold kernel waits for both (join_wgrad_streams).
// Round 4: default ON now that the side streams run below the caller's stream in priority — seven alternating rounds on one box:
// better in five, equal in two, median 5.828 against 5.900 ms per update (profiles/r04n_ab_wgrad_two_low_priority_streams.txt;
// at equal priorities round 3 had measured no gain).  VDQN_WGRAD_STREAMS=1 keeps them on one stream.
int g_wgrad_streams_override = -1;  // tools/ab_inproc.py: vdqn_debug_set_wgrad_streams (1 / 2; -1 = VDQN_WGRAD_STREAMS)
bool wgrad_two_streams(const vdqn_net* net) {
  static const bool env_on = [] { const char* e = getenv("VDQN_WGRAD_STREAMS"); return !e || atoi(e) == 2; }();
  const bool on = g_wgrad_streams_override > 0 ? g_wgrad_streams_override == 2 : env_on;
  static const bool two_stage = [] { const char* e = getenv("VDQN_WGRAD_TWO_STAGE"); return e && e[0] == '1'; }();
  return on && !two_stage && !net->cfg.deterministic;
}
}  // namespace
extern "C" void vdqn_debug_set_wgrad_streams(int v) { g_wgrad_streams_override = v; }
static int g_stem_wgrad_main_override = -1;  // tools/ab_inproc.py (0 / 1; -1 = VDQN_STEM_WGRAD_MAIN)
extern "C" void vdqn_debug_set_stem_wgrad_main(int v) { g_stem_wgrad_main_override = v; }
namespace {
hipStream_t wgrad_stream(vdqn_net* net, hipStream_t main) {
  if (!wgrad_two_streams(net)) return fork_side(net, main);
  net->wgrad_rr ^= 1;
  return net->wgrad_rr ? fork_side2(net, main) : fork_side(net, main);
}
// the first side stream (where the unfold kernel runs) waits for the weight gradients queued on the second one
void join_wgrad_streams(vdqn_net* net) {
  if (!wgrad_two_streams(net) || !net->overlap || !net->side2) return;
  hipEvent_t e = next_event(net);
  (void)hipEventRecord(e, net->side2);
  (void)hipStreamWaitEvent(net->side, e, 0);
}

// ---------------------------------------------------------------------------------------------------------
// fold / unfold kernels
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long fold_src_index(const FoldDesc& d, int co, int k) {
  if (d.kind == K_CONV1_S2D) {
    const int a = k >> 6, j = (k >> 4) & 3, ch = k & 15;
    if (ch >= 12) return -1;
    const int bh = ch / 6, bw = (ch / 3) & 1, c = ch % 3;
    const int r7 = 2 * a + bh - 1, s7 = 2 * j + bw - 1;
    if (r7 < 0 || s7 < 0) return -1;
    return ((long)(co * 3 + c) * 7 + r7) * 7 + s7;
  } else if (d.kind == K_LINEAR_PERM) {
    const int f = k / 1600, rem = k - f * 1600;
    const int hw = rem >> 6, c = rem & 63;
    return (long)co * d.ci + f * 1600 + c * 25 + hw;
  } else {
    const int tap = k / d.k_ci, c = k - tap * d.k_ci;
    const int kr = tap / d.k_s, ks = tap - kr * d.k_s;
    return (((long)co * d.ci + c) * d.r + kr) * d.s + ks;
  }
}

__device__ __forceinline__ float fold_scale(const FoldDesc& d, const float* params, const float* bnstats, int co, int raw) {
  if (!d.has_bn || raw) return 1.0f;
  return params[d.g_off + co] / sqrtf(bnstats[d.var_off + co] + kBnEps);
}

// grid: (blocks, layers, 2): z = 0 packs Wf (+bias, scale), z = 1 packs Wd.  raw = 1: BatchNorm is NOT folded
// (train-mode BatchNorm of ARCHITECTURE='basic': the convs produce the raw output, bias 0)
template <typename T>
__global__ __launch_bounds__(256) void fold_kernel(const FoldTable tab, const float* __restrict__ params, const float* __restrict__ bnstats,
                                                   unsigned char* __restrict__ packed, int with_dgrad, int raw, int first_layer) {
  const FoldDesc& d = tab.d[first_layer + blockIdx.y];
  const int which = blockIdx.z;
  if (d.tiled) return;  // fold_tile_kernel's layers
  if (which == 1 && (!with_dgrad || d.wd_off < 0)) return;
  const long stride = (long)gridDim.x * blockDim.x;
  if (which == 0) {
    T* wf = reinterpret_cast<T*>(packed + d.wf_off);
    float* bias = reinterpret_cast<float*>(packed + d.bias_off);
    float* scale = reinterpret_cast<float*>(packed + d.scale_off);
    const long total = (long)d.co_pad * d.kf;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int row = (int)(i / d.kf), k = (int)(i - (long)row * d.kf);
      float v = 0.f;
      if (row < d.co) {
        const float sc = fold_scale(d, params, bnstats, row, raw);
        const long src = fold_src_index(d, row, k);
        if (src >= 0) v = params[d.w_off + src] * sc;
        if (k == 0) {
          scale[row] = sc;
          float b = 0.f;
          if (d.has_bn) b = raw ? 0.f : params[d.b_off + row] - bnstats[d.mean_off + row] * sc;
          else if (d.has_bias) b = params[d.b_off + row];
          bias[row] = b;
        }
      } else if (k == 0) {
        scale[row] = 0.f;
        bias[row] = 0.f;
      }
      wf[i] = from_f32<T>(v);
    }
  } else {
    T* wd = reinterpret_cast<T*>(packed + d.wd_off);
    const long total = (long)d.cd_rows * d.kd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int n = (int)(i / d.kd), kk = (int)(i - (long)n * d.kd);
      const int tap = kk / d.co_pad, co = kk - tap * d.co_pad;
      float v = 0.f;
      if (co < d.co) {
        const long src = fold_src_index(d, co, tap * d.k_ci + n);
        if (src >= 0) v = params[d.w_off + src] * fold_scale(d, params, bnstats, co, raw);
      }
      wd[i] = from_f32<T>(v);
    }
  }
}

// Plain convolutions (3x3 / 1x1, ci % 64 == 0): one block packs a tile of 32 output channels x 64 input channels x all
// taps.  The OIHW source of such a tile is 32 contiguous runs of 64*taps floats (coalesced 16-byte reads, each master
// weight fetched once); the tile is transposed through LDS and written as contiguous runs into BOTH packed operands
// (Wf rows [co][tap][c], Wd rows [c][tap][co]).  The element-wise kernel above read the master weights with a stride of `taps`
// floats for Wf and of ci*taps floats for Wd: 706 MB of HBM traffic per launch for ~150 MB of algorithmic bytes (PMC).
// grid: (64x64-channel tiles over all tiled layers, 2 halves of 32 output channels)
#ifndef VDQN_FOLD_COT
#define VDQN_FOLD_COT 32
#endif
constexpr int kFoldCot = VDQN_FOLD_COT;  // output channels per fold_tile block (a build-time choice: 64 / kFoldCot blocks per 64 x 64 tile)

template <typename T, int TAPS>
__device__ __forceinline__ void fold_tile_body(const FoldDesc& d, const float* __restrict__ params, const float* __restrict__ bnstats,
                                               unsigned char* __restrict__ packed, int with_dgrad, int raw, float* sW, float* s_scale, int t, int cot_sub) {
  constexpr int COT = kFoldCot;        // output channels per block
  constexpr int RUN = 64 * TAPS;       // floats per output channel in this tile (contiguous in OIHW)
  constexpr int PITCH = RUN + 1;       // LDS row pitch: odd, so the column walk of the Wd pass spreads over the banks
  const int ci_tiles = d.ci / 64;
  const int cot = t / ci_tiles, cit = t - cot * ci_tiles;
  const int co0 = cot * 64 + cot_sub * COT, c0 = cit * 64;
  if (threadIdx.x < COT) {
    const int co = co0 + threadIdx.x;
    const float sc = co < d.co ? fold_scale(d, params, bnstats, co, raw) : 0.f;
    s_scale[threadIdx.x] = sc;
    if (cit == 0) {
      float b = 0.f;
      if (co < d.co) {
        if (d.has_bn) b = raw ? 0.f : params[d.b_off + co] - bnstats[d.mean_off + co] * sc;
        else if (d.has_bias) b = params[d.b_off + co];
      }
      reinterpret_cast<float*>(packed + d.bias_off)[co] = b;
      reinterpret_cast<float*>(packed + d.scale_off)[co] = sc;
    }
  }
  __syncthreads();
  // master weights -> LDS in source order, scaled: 16-byte loads of the contiguous [c][tap] run of every output channel
#pragma unroll 6
  for (int i = threadIdx.x; i < COT * RUN / 4; i += 256) {
    const int co_l = i / (RUN / 4), q = i - co_l * (RUN / 4);
    const int co = co0 + co_l;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (co < d.co) v = *reinterpret_cast<const float4*>(params + d.w_off + ((long)co * d.ci + c0) * TAPS + 4 * q);
    const float sc = s_scale[co_l];
    float* dst = sW + co_l * PITCH + 4 * q;
    dst[0] = v.x * sc; dst[1] = v.y * sc; dst[2] = v.z * sc; dst[3] = v.w * sc;
  }
  __syncthreads();
  // Wf rows [co][tap][c]: a thread writes EIGHT consecutive channels (16 bytes of bf16; eight lanes = the 64-channel run of one tap)
  T* wf = reinterpret_cast<T*>(packed + d.wf_off);
  constexpr int V16 = (int)(8 * sizeof(T) / 16);  // 16-byte stores per eight elements
#pragma unroll 3
  for (int i = threadIdx.x; i < COT * TAPS * 8; i += 256) {
    const int co_l = i / (TAPS * 8), rem = i - co_l * (TAPS * 8);
    const int tap = rem >> 3, c_l = (rem & 7) * 8;
    const float* src = sW + co_l * PITCH + c_l * TAPS + tap;
    T o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = from_f32<T>(src[e * TAPS]);
    T* dst = wf + (long)(co0 + co_l) * d.kf + tap * d.ci + c0 + c_l;
#pragma unroll
    for (int v = 0; v < V16; ++v) reinterpret_cast<uint4*>(dst)[v] = reinterpret_cast<const uint4*>(o8)[v];
  }
  if (with_dgrad && d.wd_off >= 0) {
    // Wd rows [c][tap][co]: a thread writes eight consecutive output channels (four lanes = this block's 32 of them)
    T* wd = reinterpret_cast<T*>(packed + d.wd_off);
#pragma unroll 3
    for (int i = threadIdx.x; i < 64 * TAPS * (COT / 8); i += 256) {
      const int co_l = (i % (COT / 8)) * 8, rem = i / (COT / 8);
      const int tap = rem % TAPS, c_l = rem / TAPS;
      const float* src = sW + co_l * PITCH + c_l * TAPS + tap;
      T o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = from_f32<T>(src[e * PITCH]);
      T* dst = wd + (long)(c0 + c_l) * d.kd + tap * d.co_pad + co0 + co_l;
#pragma unroll
      for (int v = 0; v < V16; ++v) reinterpret_cast<uint4*>(dst)[v] = reinterpret_cast<const uint4*>(o8)[v];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void fold_tile_kernel(const FoldTable tab, const float* __restrict__ params, const float* __restrict__ bnstats,
                                                        unsigned char* __restrict__ packed, int with_dgrad, int raw, int tile_first) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fold_smem[];
  float* sW = reinterpret_cast<float*>(fold_smem);  // [32][64 * taps + 1] f32, source order
  __shared__ float s_scale[32];
  int li = 0;
  for (int i = 0; i < tab.n; ++i)
    if (tab.d[i].tiled && tile_first + (int)blockIdx.x >= tab.d[i].tile_begin) li = i;
  const FoldDesc& d = tab.d[li];
  const int t = tile_first + (int)blockIdx.x - d.tile_begin;
  if (d.r * d.s == 9) fold_tile_body<T, 9>(d, params, bnstats, packed, with_dgrad, raw, sW, s_scale, t, (int)blockIdx.y);
  else fold_tile_body<T, 1>(d, params, bnstats, packed, with_dgrad, raw, sW, s_scale, t, (int)blockIdx.y);
}

// grid: (max co, layers of the stage): one block per output channel
// raw = 1: the weights were packed without BatchNorm folding; the BatchNorm parameter gradients were already written
// by the train-mode BatchNorm backward
__global__ __launch_bounds__(256) void unfold_kernel(const FoldTable tab, const PartTable pt, int first_layer, const float* __restrict__ params,
                                                     const float* __restrict__ bnstats, const unsigned char* __restrict__ bwd,
                                                     float* __restrict__ grads, int raw) {
  const FoldDesc& d = tab.d[first_layer + blockIdx.y];
  const int co = blockIdx.x;
  if (co >= d.co) return;
  const float* dw = reinterpret_cast<const float*>(bwd + d.dw_off) + (long)co * d.kf;
  const float* db = reinterpret_cast<const float*>(bwd + d.db_off);
  const int li = first_layer + blockIdx.y;
  float dbsum = 0.f;
  if (pt.tiles[li] > 0) {
    const float* part = reinterpret_cast<const float*>(bwd + pt.off[li]);
    const int per_tile = pt.groups[li];
    const int total = pt.tiles[li] * per_tile;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int t = i / per_tile, g = i - t * per_tile;
      dbsum += part[(long)t * pt.ld[li] + g * pt.gstride[li] + co];
    }
  }
  float rstd = 1.f, sc = 1.f;
  if (d.has_bn && !raw) {
    rstd = 1.0f / sqrtf(bnstats[d.var_off + co] + kBnEps);
    sc = params[d.g_off + co] * rstd;
  }
  float dot = 0.f;
  __shared__ __attribute__((aligned(16))) float s_row[4608];  // one packed-layout dW' row of a plain convolution (<= 9 taps x 512 channels)
  if (d.tiled && d.kf <= 4608) {
    // packed row -> LDS (coalesced), then the OIHW row of the gradient and of the master weights is walked in ITS order
    // (coalesced global accesses; the [tap][c] -> [c][tap] permutation happens on the LDS read)
    const int taps = d.r * d.s;
    const long base = d.w_off + (long)co * d.kf;
    if ((((uintptr_t)dw | (uintptr_t)(params + base) | (uintptr_t)(grads + base)) & 15) == 0 && (d.kf & 3) == 0) {
      // 16-byte accesses on both sides of the permutation (a stage's unfold moves up to 100 MB)
      for (int k = threadIdx.x; k < d.kf / 4; k += 256) reinterpret_cast<float4*>(s_row)[k] = reinterpret_cast<const float4*>(dw)[k];
      __syncthreads();
      for (int i4 = threadIdx.x; i4 < d.kf / 4; i4 += 256) {
        float g[4];
        int c = (4 * i4) / taps, tap = 4 * i4 - c * taps;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[e] = s_row[tap * d.ci + c];
          if (++tap == taps) { tap = 0; ++c; }
        }
        const float4 w = reinterpret_cast<const float4*>(params + base)[i4];
        reinterpret_cast<float4*>(grads + base)[i4] = make_float4(g[0] * sc, g[1] * sc, g[2] * sc, g[3] * sc);
        dot += g[0] * w.x + g[1] * w.y + g[2] * w.z + g[3] * w.w;
      }
    } else {
      for (int k = threadIdx.x; k < d.kf; k += 256) s_row[k] = dw[k];
      __syncthreads();
      for (int i = threadIdx.x; i < d.kf; i += 256) {
        const int c = i / taps, tap = i - c * taps;
        const float g = s_row[tap * d.ci + c];
        grads[base + i] = g * sc;
        dot += g * params[base + i];
      }
    }
  } else {
    for (int k = threadIdx.x; k < d.kf; k += 256) {
      const long src = fold_src_index(d, co, k);
      if (src >= 0) {
        const float g = dw[k];
        grads[d.w_off + src] = g * sc;
        dot += g * params[d.w_off + src];
      }
    }
  }
  __shared__ float red[8];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dot += __shfl_down(dot, o, 64);
    dbsum += __shfl_down(dbsum, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = dot;
    red[4 + (threadIdx.x >> 6)] = dbsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float dbp = pt.tiles[li] > 0 ? (red[4] + red[5] + red[6] + red[7]) : db[co];
    if (d.has_bn) {
      if (!raw) {
        grads[d.g_off + co] = rstd * (tot - bnstats[d.mean_off + co] * dbp);
        grads[d.b_off + co] = dbp;
      }
    } else if (d.has_bias) {
      grads[d.b_off + co] = dbp;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// table construction
// ---------------------------------------------------------------------------------------------------------
void add_param(vdqn_net* net, const std::string& name, int64_t off, std::vector<int> shape, int kind, int stage) {
  vdqn_param_info pi;
  memset(&pi, 0, sizeof(pi));
  snprintf(pi.name, sizeof(pi.name), "%s", name.c_str());
  pi.offset = off;
  pi.ndim = (int)shape.size();
  int64_t n = 1;
  for (size_t i = 0; i < shape.size(); ++i) {
    pi.shape[i] = shape[i];
    n *= shape[i];
  }
  pi.numel = n;
  pi.kind = kind;
  pi.param_id = -1;
  pi.stage = stage;
  net->params.push_back(pi);
}

Layer make_conv(const std::string& name, const std::string& bn, int co, int ci, int k, int stride, int pad, int hi, int stage) {
  Layer L;
  L.name = name;
  L.bn_name = bn;
  L.kind = K_CONV;
  L.co = co; L.ci = ci; L.r = k; L.s = k; L.stride = stride; L.pad = pad;
  L.has_bn = !bn.empty();
  L.has_bias = bn.empty();
  L.k_ci = ci; L.k_r = k; L.k_s = k; L.pix_stride = ci;
  L.co_pad = (co + 63) / 64 * 64;
  L.hi = hi; L.wi = hi;
  L.ho = (hi + 2 * pad - k) / stride + 1;
  L.wo = L.ho;
  L.per_sample = 0;
  L.stage = stage;
  L.has_dgrad = 1;
  return L;
}

Layer make_linear(const std::string& name, int out_f, int in_f, int stage, bool perm) {
  Layer L;
  L.name = name;
  L.bn_name = "";
  L.kind = perm ? K_LINEAR_PERM : K_LINEAR;
  L.co = out_f; L.ci = in_f; L.r = 1; L.s = 1; L.stride = 1; L.pad = 0;
  L.has_bn = 0;
  L.has_bias = 1;
  L.k_ci = in_f; L.k_r = 1; L.k_s = 1; L.pix_stride = in_f;
  L.co_pad = (out_f + 63) / 64 * 64;
  L.hi = L.wi = L.ho = L.wo = 1;
  L.per_sample = 1;
  L.stage = stage;
  L.has_dgrad = 1;
  return L;
}

void build_layers(vdqn_net* net) {
  const int F = net->cfg.num_frames;
  std::vector<Layer> fwd;  // forward order
  {
    Layer L = make_conv("resnet.conv1", "resnet.bn1", 64, 3, 7, 2, 3, 224, 2);
    L.kind = K_CONV1_S2D;
    L.k_ci = 64; L.k_r = 4; L.k_s = 1; L.pix_stride = 16;
    L.hi = L.wi = 115;  // packed space-to-depth operand
    L.ho = L.wo = 112;
    L.has_dgrad = 0;
    fwd.push_back(L);
  }
  int inpl = 64, sp = 56;
  for (int li = 1; li <= 4; ++li) {
    const int planes = 64 << (li - 1);
    const int stage = li == 4 ? 0 : (li == 3 ? 1 : 2);
    for (int bi = 0; bi < 2; ++bi) {
      const int stride = (li > 1 && bi == 0) ? 2 : 1;
      char pfx[64];
      snprintf(pfx, sizeof(pfx), "resnet.layer%d.%d", li, bi);
      const std::string p(pfx);
      fwd.push_back(make_conv(p + ".conv1", p + ".bn1", planes, inpl, 3, stride, 1, sp, stage));
      const int sp_out = sp / stride;
      fwd.push_back(make_conv(p + ".conv2", p + ".bn2", planes, planes, 3, 1, 1, sp_out, stage));
      if (stride != 1 || inpl != planes) fwd.push_back(make_conv(p + ".downsample.0", p + ".downsample.1", planes, inpl, 1, stride, 0, sp, stage));
      inpl = planes;
      sp = sp_out;
    }
  }
  if (!net->basic()) {  // archs/HabitatDQNMultiAction.py:27-31
    fwd.push_back(make_conv("features.8", "", 64, 512, 3, 1, 0, 7, 0));
    fwd.push_back(make_linear("top.0", 512, 1600 * F, 0, true));
    fwd.push_back(make_linear("top.2", 256, 512, 0, false));
    fwd.push_back(make_linear("top.4", net->cfg.action_dim * net->cfg.num_classes, 256, 0, false));
  } else {  // :32-34: global average pool, then one Linear over the F concatenated 512-vectors
    fwd.push_back(make_linear("top", net->cfg.action_dim * net->cfg.num_classes, 512 * F, 0, false));
  }

  // store layers ordered by backward stage (stable), so each stage's gradients are one contiguous range
  net->layers.clear();
  for (int st = 0; st < 3; ++st) {
    net->layer_stage_first[st] = (int)net->layers.size();
    for (auto& L : fwd)
      if (L.stage == st) net->layers.push_back(L);
    net->layer_stage_count[st] = (int)net->layers.size() - net->layer_stage_first[st];
  }

  auto find = [&](const std::string& n) {
    for (size_t i = 0; i < net->layers.size(); ++i)
      if (net->layers[i].name == n) return (int)i;
    return -1;
  };
  net->l_conv1 = find("resnet.conv1");
  net->l_f8 = find("features.8");
  net->l_top0 = find("top.0");
  net->l_top2 = find("top.2");
  net->l_top4 = net->basic() ? find("top") : find("top.4");
  for (int b = 0; b < 8; ++b) {
    char pfx[64];
    snprintf(pfx, sizeof(pfx), "resnet.layer%d.%d", b / 2 + 1, b % 2);
    net->l_b_conv1[b] = find(std::string(pfx) + ".conv1");
    net->l_b_conv2[b] = find(std::string(pfx) + ".conv2");
    net->l_b_ds[b] = find(std::string(pfx) + ".downsample.0");
  }

  // flat offsets: trainable parameters grouped by stage, then the frozen resnet.fc
  int64_t poff = 0, soff = 0, pk = 0, dwoff = 0;
  const int esz = net->esz;
  for (int st = 0; st < 3; ++st) {
    net->stage_begin[st] = poff;
    for (int i = net->layer_stage_first[st]; i < net->layer_stage_first[st] + net->layer_stage_count[st]; ++i) {
      Layer& L = net->layers[i];
      L.w_off = poff;
      if (L.kind == K_LINEAR || L.kind == K_LINEAR_PERM) add_param(net, L.name + ".weight", poff, {L.co, L.ci}, 0, st);
      else add_param(net, L.name + ".weight", poff, {L.co, L.ci, L.r, L.s}, 0, st);
      poff += (int64_t)L.co * L.ci * L.r * L.s;
      poff = (poff + 3) / 4 * 4;  // keep every tensor 16-byte aligned
      L.g_off = L.b_off = L.mean_off = L.var_off = -1;
      if (L.has_bn) {
        L.g_off = poff;
        add_param(net, L.bn_name + ".weight", poff, {L.co}, 0, st);
        poff += L.co;
        L.b_off = poff;
        add_param(net, L.bn_name + ".bias", poff, {L.co}, 0, st);
        poff += L.co;
        L.mean_off = soff;
        add_param(net, L.bn_name + ".running_mean", soff, {L.co}, 2, st);
        soff += L.co;
        L.var_off = soff;
        add_param(net, L.bn_name + ".running_var", soff, {L.co}, 3, st);
        soff += L.co;
      } else if (L.has_bias) {
        L.b_off = poff;
        add_param(net, L.name + ".bias", poff, {L.co}, 0, st);
        poff += L.co;
        poff = (poff + 3) / 4 * 4;
      }
      // packed weights
      L.wf_off = pk;
      pk = align_up(pk + (int64_t)L.co_pad * L.kf() * esz);
      if (L.has_dgrad) {
        L.wd_off = pk;
        pk = align_up(pk + (int64_t)L.k_ci * L.kd() * esz);
      } else {
        L.wd_off = -1;
      }
      L.bias_off = pk;
      pk = align_up(pk + (int64_t)L.co_pad * 4);
      L.scale_off = pk;
      pk = align_up(pk + (int64_t)L.co_pad * 4);
      // f32 gradient accumulators
      L.dw_off = dwoff;
      dwoff = align_up(dwoff + (int64_t)L.co_pad * L.kf() * 4);
      L.db_off = dwoff;
      dwoff = align_up(dwoff + (int64_t)L.co_pad * 4);
    }
    net->stage_end[st] = poff;
  }
  net->trainable_numel = poff;
  add_param(net, "resnet.fc.weight", poff, {1000, 512}, 1, -1);
  poff += 1000 * 512;
  add_param(net, "resnet.fc.bias", poff, {1000}, 1, -1);
  poff += 1000;
  net->params_numel = poff;
  net->bnstats_numel = soff;
  net->packed_bytes = pk;
  net->dw_bytes = dwoff;

  // reference model.parameters() order -> param_id (Adam state_dict ids)
  {
    std::vector<std::string> order;
    for (auto& L : fwd) {
      if (L.name.rfind("resnet.", 0) != 0) continue;
      order.push_back(L.name + ".weight");
      order.push_back(L.bn_name + ".weight");
      order.push_back(L.bn_name + ".bias");
    }
    order.push_back("resnet.fc.weight");
    order.push_back("resnet.fc.bias");
    if (!net->basic()) {
      for (const char* n : {"features.8", "top.0", "top.2", "top.4"}) {
        order.push_back(std::string(n) + ".weight");
        order.push_back(std::string(n) + ".bias");
      }
    } else {
      order.push_back("top.weight");
      order.push_back("top.bias");
    }
    for (auto& pi : net->params)
      for (size_t i = 0; i < order.size(); ++i)
        if (order[i] == pi.name) pi.param_id = (int)i;
  }

  // device-side descriptors
  net->fold.n = (int)net->layers.size();
  for (size_t i = 0; i < net->layers.size(); ++i) {
    const Layer& L = net->layers[i];
    FoldDesc& d = net->fold.d[i];
    d.w_off = L.w_off; d.g_off = L.g_off; d.b_off = L.b_off; d.mean_off = L.mean_off; d.var_off = L.var_off;
    d.wf_off = L.wf_off; d.wd_off = L.wd_off; d.bias_off = L.bias_off; d.scale_off = L.scale_off;
    d.dw_off = L.dw_off; d.db_off = L.db_off;
    d.co = L.co; d.ci = L.ci; d.r = L.r; d.s = L.s; d.kind = L.kind; d.co_pad = L.co_pad; d.kf = L.kf();
    d.k_ci = L.k_ci; d.k_s = L.k_s; d.cd_rows = L.k_ci; d.kd = L.kd(); d.has_bn = L.has_bn; d.has_bias = L.has_bias;
    d.tiled = (L.kind == K_CONV && L.ci % 64 == 0 && (L.r * L.s == 9 || L.r * L.s == 1)) ? 1 : 0;
    d.tile_begin = 0;
  }
  net->fold.n_tiles = 0;
  for (int i = 0; i < net->fold.n; ++i) {
    FoldDesc& d = net->fold.d[i];
    if (!d.tiled) continue;
    d.tile_begin = net->fold.n_tiles;
    net->fold.n_tiles += (d.co_pad / 64) * (d.ci / 64);
  }
}

ActLayout act_layout(const vdqn_net* net, int n_samples) {
  const int64_t F = net->cfg.num_frames, n = (int64_t)n_samples * F, e = net->esz;
  ActLayout L;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    const int64_t o = off;
    off = align_up(off + bytes);
    return o;
  };
  L.t_in = take(n * 115 * 115 * 16 * e);
  L.c1 = net->basic() ? take(n * 112 * 112 * 64 * e) : -1;  // extra_capacity: conv1 + max-pool are one kernel, c1 never exists
  L.pool = take(n * 56 * 56 * 64 * e);
  L.idx = take(n * 56 * 56 * 64);
  for (int b = 0; b < 8; ++b) {
    const int li = b / 2;
    const int64_t planes = 64 << li, sp = 56 >> li;
    const int64_t sz = n * sp * sp * planes * e;
    L.h[b] = take(sz);
    L.o[b] = take(sz);
    L.ds[b] = (b % 2 == 0 && li > 0) ? take(sz) : -1;
  }
  L.f8 = take(n * 25 * 64 * e);
  L.l0 = take((int64_t)n_samples * 512 * e);
  L.l1 = take((int64_t)n_samples * 256 * e);
  L.q = take((int64_t)n_samples * 64 * e);
  L.qf = take((int64_t)n_samples * 64 * 4);
  L.avg = L.r_c1 = L.bnw_begin = L.bn_sync = -1;
  L.bnw_bytes = 0;
  for (int b = 0; b < 8; ++b) L.r_h[b] = L.r_o[b] = L.r_ds[b] = -1;
  for (int i = 0; i < kMaxLayers; ++i) L.bnw[i] = -1;
  if (net->basic()) {
    L.avg = take(n * 512 * e);
    L.r_c1 = take(n * 112 * 112 * 64 * e);
    for (int b = 0; b < 8; ++b) {
      const int li = b / 2;
      const int64_t planes = 64 << li, sp = 56 >> li;
      const int64_t sz = n * sp * sp * planes * e;
      L.r_h[b] = take(sz);
      L.r_o[b] = take(sz);
      L.r_ds[b] = (b % 2 == 0 && li > 0) ? take(sz) : -1;
    }
    L.bnw_begin = off;
    for (size_t i = 0; i < net->layers.size(); ++i)
      if (net->layers[i].has_bn) L.bnw[i] = take((int64_t)2 * F * 6 * net->layers[i].co * 4);
    L.bnw_bytes = off - L.bnw_begin;
    L.bn_sync = take((int64_t)2 * F * 2 * 512 * 4);  // packed sums of one layer (SyncBN scratch)
    if (net->cfg.deterministic) {  // ordered two-stage statistic sums: the largest per-block partial array of any layer and call shape
      for (const Layer& ly : net->layers) {
        if (!ly.has_bn) continue;
        for (int halves = 1; halves <= 2; ++halves) {
          if (n % halves || (n / halves) % F) continue;
          L.bn_det_bytes = std::max(L.bn_det_bytes, vdqn_bn_train_workspace_bytes((int32_t)n, ly.ho * ly.wo, ly.co, (int32_t)F, (int32_t)(n / halves)));
        }
      }
      L.bn_det = take(L.bn_det_bytes);
    }
  }
  L.total = off;
  return L;
}

vdqn_wgrad_args wgrad_shape_args(const vdqn_net* net, const Layer& L, int n_units);
int64_t wgrad_max_imgs(const vdqn_net* net, const Layer& L);

bool wgrad_two_stage();

BwdLayout bwd_layout(const vdqn_net* net, int n_samples) {
  const int64_t F = net->cfg.num_frames, n = (int64_t)n_samples * F, e = net->esz;
  BwdLayout L;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    const int64_t o = off;
    off = align_up(off + bytes);
    return o;
  };
  L.zero_begin = 0;
  take(net->dw_bytes);
  L.zero_bytes = off;
  L.dq = take((int64_t)n_samples * 64 * e);
  L.g_l1 = take((int64_t)n_samples * 256 * e);
  L.g_l0 = take((int64_t)n_samples * 512 * e);
  L.g_f8 = take(n * 25 * 64 * e);
  for (int b = 0; b < 8; ++b) {
    const int li = b / 2;
    const int64_t planes = 64 << li, sp = 56 >> li;
    const int64_t sz = n * sp * sp * planes * e;
    L.g_o[b] = take(sz);
    L.g_h[b] = take(sz);
    // gradient of the downsample branch w.r.t. the block input (input geometry of the block)
    L.dsg[b] = (b % 2 == 0 && li > 0) ? take(n * (sp * 2) * (sp * 2) * (planes / 2) * e) : -1;
  }
  L.g_pool = take(n * 56 * 56 * 64 * e);
  L.g_c1 = take(n * 112 * 112 * 64 * e);
  auto tiles = [](int64_t rows) { return (rows + 127) / 128; };
  auto tiles32 = [](int64_t rows) { return (rows + 31) / 32; };  // the head's skinny kernels write one entry per 32 rows (skinny.hip)
  L.p_l1 = take(tiles32(n_samples) * 256 * 4);
  L.p_l0 = take(tiles32(n_samples) * 512 * 4);
  L.p_f8 = take(tiles32(n_samples) * 1600 * F * 4);
  L.p_pool = take(tiles(n * 56 * 56) * 64 * 4);
  for (int b = 0; b < 8; ++b) {
    const int64_t planes = 64 << (b / 2), sp = 56 >> (b / 2);
    L.p_o[b] = take((tiles(n * sp * sp) + 4) * planes * 4);  // +4: stride-2 dgrad rounds tiles per parity class
    L.p_h[b] = take(tiles(n * sp * sp) * planes * 4);
  }
  L.det_ws = -1;
  L.det_ws_bytes = 0;
  if (net->cfg.deterministic || wgrad_two_stage()) {
    for (const Layer& ly : net->layers) {
      const int64_t units = ly.per_sample ? n_samples : n;
      const int64_t mx = wgrad_max_imgs(net, ly);
      if (mx < 1) continue;  // run_wgrad reports it
      const vdqn_wgrad_args wa = wgrad_shape_args(net, ly, (int)std::min(units, mx));
      L.det_ws_bytes = std::max(L.det_ws_bytes, vdqn_conv2d_wgrad_workspace_bytes(&wa));
    }
    L.det_ws = take(L.det_ws_bytes);
  }
  L.g_avg = -1;
  for (int b = 0; b < 8; ++b) L.g_or[b] = L.g_dsr[b] = -1;
  if (net->basic()) {
    L.g_avg = take(n * 512 * e);
    for (int b = 0; b < 8; ++b) {
      const int64_t planes = 64 << (b / 2), sp = 56 >> (b / 2);
      L.g_or[b] = take(n * sp * sp * planes * e);
      if (b % 2 == 0 && b > 0) L.g_dsr[b] = take(n * sp * sp * planes * e);
    }
  }
  L.total = off;
  return L;
}

// ---------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------
// per-layer rows in the launch profiler (VDQN_PROFILE_LAYERS=1): "<kernel>|<layer> n<images>"
void prof_layer(const Layer& L, int n_units) {
  static const bool on = [] { const char* e = getenv("VDQN_PROFILE_LAYERS"); return e && e[0] == '1'; }();
  if (!on) return;
  static thread_local char buf[64];
  const char* nm = L.name.c_str();
  if (strncmp(nm, "resnet.", 7) == 0) nm += 7;
  snprintf(buf, sizeof(buf), "%s n%d", nm, n_units);
  g_prof_suffix = buf;
}

// the geometry part of a layer's weight-gradient call (no pointers): what vdqn_conv2d_wgrad_workspace_bytes needs
vdqn_wgrad_args wgrad_shape_args(const vdqn_net* net, const Layer& L, int n_units) {
  vdqn_wgrad_args a;
  memset(&a, 0, sizeof(a));
  a.n_img = n_units; a.hi = L.hi; a.wi = L.wi; a.ci = L.k_ci; a.pix_stride = L.pix_stride;
  a.ho = L.ho; a.wo = L.wo; a.co = L.co_pad; a.ldg = L.co_pad;
  a.r = L.k_r; a.s = L.k_s;
  a.stride = L.kind == K_CONV1_S2D ? 1 : L.stride;
  a.pad = L.kind == K_CONV1_S2D ? 0 : L.pad;
  a.splitk = 0; a.dtype = net->cfg.dtype;
  return a;
}
// images one vdqn_conv2d_wgrad call can take for layer L (< 2^24 output pixels, < 2 GiB per operand: 32-bit buffer offsets)
int64_t wgrad_max_imgs(const vdqn_net* net, const Layer& L) {
  const int64_t esz = net->esz;
  const int64_t pix = (int64_t)L.ho * L.wo, gy_img = pix * L.co_pad * esz, x_img = (int64_t)L.hi * L.wi * L.pix_stride * esz;
  int64_t m = ((1ll << 24) - 1) / pix;
  m = std::min(m, (int64_t)0x7ffffffell / gy_img);
  m = std::min(m, (int64_t)0x7ffffffell / x_img);
  return m;
}

// VDQN_FUSE_DS: the 1x1 downsample of a stride-2 BasicBlock rides in its sibling 3x3's launches.  Bit 0 (default on): backward —
// extra K-steps of the 3x3's stride-2 data gradient, the shortcut gradient never exists (0.27 instead of 0.42 ms per update);
// bit 1 (default on since round 5, bf16 engines): forward — second output of one launch, three launches fewer per pass,
// bit-identical outputs: the persistent plane-window kernel (win9s.hip, win9sp_kernel) runs the 1x1 as extra K-steps on the P00
// window behind the 3x3's epilogue.  (On the generic kernel — f32 engines, odd-sized inputs — the short sibling tiles between the
// long ones cost more than their own launch did: bit 2 forces the fused form there too.)
// VDQN_WGRAD_TWO_STAGE=1: the split-K weight-gradient partials as plain stores into per-split copies + one ordered reduce kernel
// per layer (the deterministic mode's path, include/vdqn.h vdqn_wgrad_args.workspace) also in the default mode — instead of
// ~25-50 MB of f32 atomics per launch at the ~1.3 TB/s the memory side sustains for them
bool wgrad_two_stage() {
  static const bool on = [] { const char* e = getenv("VDQN_WGRAD_TWO_STAGE"); return e && e[0] == '1'; }();
  return on;
}

int fuse_ds_mask() {
  static const int m = [] { const char* e = getenv("VDQN_FUSE_DS"); return e ? atoi(e) : 3; }();
  return m;
}
bool fuse_ds() { return (fuse_ds_mask() & 1) != 0; }
bool fuse_ds_fwd(int dtype) { return (fuse_ds_mask() & 2) != 0 && (dtype == VDQN_BF16 || (fuse_ds_mask() & 4) != 0); }

int run_conv(const vdqn_net* net, const Layer& L, const unsigned char* packed, const void* in, void* out, int n_units, const void* resid,
             int relu, float* out_f32, hipStream_t st, const Layer* sib = nullptr, void* sib_out = nullptr) {
  vdqn_conv_args a;
  memset(&a, 0, sizeof(a));
  a.in = in;
  a.wt = packed + L.wf_off;
  a.bias = reinterpret_cast<const float*>(packed + L.bias_off);
  a.resid = resid;
  a.mask = nullptr;
  a.out = out;
  a.out_f32 = out_f32;
  a.n_img = n_units; a.hi = L.hi; a.wi = L.wi; a.ci = L.k_ci; a.pix_stride = L.pix_stride;
  a.ho = L.ho; a.wo = L.wo; a.co = L.co_pad; a.ldo = L.co_pad;
  a.r = L.k_r; a.s = L.k_s;
  a.stride = L.kind == K_CONV1_S2D ? 1 : L.stride;
  a.pad = L.kind == K_CONV1_S2D ? 0 : L.pad;
  a.mode = 0; a.relu = relu; a.dtype = net->cfg.dtype;
  g_prof_alg_flops = 2.0 * n_units * L.ho * L.wo * (double)L.co * L.ci * L.r * L.s;
  if (sib) {  // the block's 1x1 / stride-2 downsample (BatchNorm folded, no ReLU): second output of the same launch
    a.wt2 = packed + sib->wf_off;
    a.bias2 = reinterpret_cast<const float*>(packed + sib->bias_off);
    a.out2 = sib_out;
    a.co2 = sib->co_pad; a.ldo2 = sib->co_pad; a.relu2 = 0;
    g_prof_alg_flops += 2.0 * n_units * sib->ho * sib->wo * (double)sib->co * sib->ci;
  }
  prof_layer(L, n_units);
  return vdqn_conv2d(&a, st);
}

// data gradient: gx = (dgrad(gy) + resid) masked by (mask > 0)
int run_dgrad(const vdqn_net* net, const Layer& L, const unsigned char* packed, const void* gy, void* gx, int n_units, const void* resid,
              const void* mask, hipStream_t st, void* colsum_part = nullptr, const Layer* sib = nullptr, const void* sib_gy = nullptr,
              int* part_rows = nullptr) {
  vdqn_conv_args a;
  memset(&a, 0, sizeof(a));
  a.in = gy;
  a.wt = packed + L.wd_off;
  a.bias = nullptr;
  a.resid = resid;
  a.mask = mask;
  a.out = gx;
  a.colsum_part = reinterpret_cast<float*>(colsum_part);
  a.n_img = n_units; a.hi = L.ho; a.wi = L.wo; a.ci = L.co_pad; a.pix_stride = L.co_pad;
  a.ho = L.hi; a.wo = L.wi; a.co = L.k_ci; a.ldo = L.k_ci;
  a.r = L.k_r; a.s = L.k_s; a.stride = L.stride; a.pad = L.pad;
  a.mode = 1; a.relu = 0; a.dtype = net->cfg.dtype;
  g_prof_alg_flops = 2.0 * n_units * L.ho * L.wo * (double)L.co * L.ci * L.r * L.s;
  if (sib) {  // + the data gradient of the block's 1x1 / stride-2 downsample, accumulated in the same tiles
    a.in2 = sib_gy;
    a.wt2 = packed + sib->wd_off;
    a.ci2 = sib->co_pad;
    g_prof_alg_flops += 2.0 * n_units * sib->ho * sib->wo * (double)sib->co * sib->ci;
  }
  if (part_rows) {  // rows of gx one entry of colsum_part covers: the kernel vdqn_conv2d picks for this call decides
    const int kind = vdqn_skinny_kind(&a);
    *part_rows = kind ? vdqn_skinny_part_rows(kind == 2) : 128;
  }
  prof_layer(L, n_units);
  return vdqn_conv2d(&a, st);
}

int run_wgrad(const vdqn_net* net, const Layer& L, unsigned char* bwd, const void* gy, const void* x, int n_units, hipStream_t st,
              bool colsum_kernel = false) {
  vdqn_wgrad_args a = wgrad_shape_args(net, L, n_units);
  a.gy = gy;
  a.x = x;
  a.dw = reinterpret_cast<float*>(bwd + L.dw_off);
  a.dbias = colsum_kernel ? reinterpret_cast<float*>(bwd + L.db_off) : nullptr;  // else: dgrad-epilogue partials
  if (net->cfg.deterministic || wgrad_two_stage()) {  // all weight gradients of an update run on ONE stream, so they can share the workspace
    const BwdLayout W = bwd_layout(net, net->bwd_samples);
    a.workspace = bwd + W.det_ws;
    a.workspace_bytes = W.det_ws_bytes;
  }
  // vdqn_conv2d_wgrad addresses < 2^24 output pixels and < 2 GiB per operand (32-bit buffer offsets): larger batches
  // (conv1 at more than 1337 bf16 / 668 f32 frames, e.g. 12-view samples at batch 256) go through it in image ranges —
  // dw and dbias accumulate, so the ranges simply add up.
  const int64_t esz = net->esz;
  const int64_t pix = (int64_t)L.ho * L.wo, gy_img = pix * L.co_pad * esz, x_img = (int64_t)L.hi * L.wi * L.pix_stride * esz;
  const int64_t max_imgs = wgrad_max_imgs(net, L);
  if (max_imgs < 1) {
    vdqn_set_error("run_wgrad: one image of layer %s exceeds the weight-gradient kernel's addressing range", L.name.c_str());
    return VDQN_ERR_INVALID;
  }
  for (int64_t i0 = 0; i0 < n_units; i0 += max_imgs) {
    const int n_chunk = (int)std::min<int64_t>(max_imgs, n_units - i0);
    a.gy = (const unsigned char*)gy + i0 * gy_img;
    a.x = (const unsigned char*)x + i0 * x_img;
    a.n_img = n_chunk;
    g_prof_alg_flops = 2.0 * n_chunk * L.ho * L.wo * (double)L.co * L.ci * L.r * L.s;
    prof_layer(L, n_chunk);
    const int rc = vdqn_conv2d_wgrad(&a, st);
    if (rc != VDQN_OK) return rc;
  }
  return VDQN_OK;
}

#define RC(x)                     \
  do {                            \
    int rc_ = (x);                \
    if (rc_ != VDQN_OK) return rc_; \
  } while (0)

// forward over n_samples samples whose packed input already sits at `t_in`
int forward_impl(const vdqn_net* net, const unsigned char* packed, const void* t_in, int n_samples, unsigned char* acts, const ActLayout& A,
                 hipStream_t st, bool trunk_only = false, int grad_samples = -1) {
  const int n = n_samples * net->cfg.num_frames;
  const int dt = net->cfg.dtype;
  // grad_samples >= 0: only the first grad_samples samples will see a backward pass — the stem skips the arg-max bytes of the
  // max-pool for the rest (the s' rows and the target pass of a TD update: 2/3 of its frames; VDQN_STEM_NOIDX=0 writes them all)
  static const bool stem_noidx = [] { const char* e = getenv("VDQN_STEM_NOIDX"); return !(e && e[0] == '0'); }();
  const int n_idx = (grad_samples >= 0 && stem_noidx) ? (grad_samples * net->cfg.num_frames < n ? grad_samples * net->cfg.num_frames : n) : n;
  if (A.c1 >= 0) {  // 'basic' eval path keeps the separate kernels (its train path needs the raw conv output anyway)
    RC(run_conv(net, net->layers[net->l_conv1], packed, t_in, acts + A.c1, n, nullptr, 1, nullptr, st));
    RC(vdqn_maxpool_fwd(acts + A.c1, acts + A.pool, acts + A.idx, n, 112, 112, 64, dt, st));
  } else {
    const Layer& L1 = net->layers[net->l_conv1];
    prof_layer(L1, n);
    RC(vdqn_stem_conv_pool_n(t_in, packed + L1.wf_off, reinterpret_cast<const float*>(packed + L1.bias_off), acts + A.pool, acts + A.idx, n, n_idx, dt, st));
  }
  const unsigned char* x = acts + A.pool;
  for (int b = 0; b < 8; ++b) {
    const Layer& c1 = net->layers[net->l_b_conv1[b]];
    const Layer& c2 = net->layers[net->l_b_conv2[b]];
    const void* identity = x;
    if (net->l_b_ds[b] >= 0 && fuse_ds_fwd(dt)) {  // stride-2 block: conv1 and the downsample read the same pixels — one launch
      RC(run_conv(net, c1, packed, x, acts + A.h[b], n, nullptr, 1, nullptr, st, &net->layers[net->l_b_ds[b]], acts + A.ds[b]));
      identity = acts + A.ds[b];
    } else {
      RC(run_conv(net, c1, packed, x, acts + A.h[b], n, nullptr, 1, nullptr, st));
      if (net->l_b_ds[b] >= 0) {
        RC(run_conv(net, net->layers[net->l_b_ds[b]], packed, x, acts + A.ds[b], n, nullptr, 0, nullptr, st));
        identity = acts + A.ds[b];
      }
    }
    RC(run_conv(net, c2, packed, acts + A.h[b], acts + A.o[b], n, identity, 1, nullptr, st));
    x = acts + A.o[b];
  }
  if (trunk_only) return VDQN_OK;  // the 512 x 7 x 7 features are in o7
  if (net->basic()) {
    RC(vdqn_avgpool_fwd(x, acts + A.avg, n, 49, 512, dt, st));
    RC(run_conv(net, net->layers[net->l_top4], packed, acts + A.avg, acts + A.q, n_samples, nullptr, 0, reinterpret_cast<float*>(acts + A.qf), st));
    return VDQN_OK;
  }
  RC(run_conv(net, net->layers[net->l_f8], packed, x, acts + A.f8, n, nullptr, 1, nullptr, st));
  RC(run_conv(net, net->layers[net->l_top0], packed, acts + A.f8, acts + A.l0, n_samples, nullptr, 1, nullptr, st));
  RC(run_conv(net, net->layers[net->l_top2], packed, acts + A.l0, acts + A.l1, n_samples, nullptr, 1, nullptr, st));
  RC(run_conv(net, net->layers[net->l_top4], packed, acts + A.l1, acts + A.q, n_samples, nullptr, 0, reinterpret_cast<float*>(acts + A.qf), st));
  return VDQN_OK;
}

// train-mode BatchNorm of layer L over the raw conv output y: z = relu?(bn(y) (+ resid)); updates the running statistics
int run_bn(const vdqn_net* net, int li, const float* params, float* bnstats, unsigned char* acts, const ActLayout& A, const void* y,
           const void* resid, void* z, int n_img, int iph, int relu, hipStream_t st) {
  const Layer& L = net->layers[li];
  BnSync sy = net->bn_sync;
  sy.scratch = reinterpret_cast<float*>(acts + A.bn_sync);
  if (A.bn_det >= 0) {
    sy.det_ws = reinterpret_cast<float*>(acts + A.bn_det);
    sy.det_ws_bytes = A.bn_det_bytes;
  }
  return vdqn_bn_train_fwd_impl(y, resid, z, params + L.g_off, params + L.b_off, bnstats + L.mean_off, bnstats + L.var_off,
                                reinterpret_cast<float*>(acts + A.bnw[li]), n_img, L.ho * L.wo, L.co, net->cfg.num_frames, iph, relu, kBnMomentum,
                                kBnEps, net->cfg.dtype, st, &sy);
}

// ARCHITECTURE='basic' in train mode (archs/HabitatDQNMultiAction.py:37-40 leaves the ResNet in train mode): forward over
// n_samples samples = n_samples/halves per model call, batch statistics per (call, frame slot), raw conv outputs kept for
// the backward.  `packed` must hold the un-folded weights (vdqn_net_pack_weights flag 2).
int forward_train_impl(const vdqn_net* net, const unsigned char* packed, const float* params, float* bnstats, const void* t_in, int n_samples,
                       int halves, unsigned char* acts, const ActLayout& A, hipStream_t st) {
  const int F = net->cfg.num_frames, n = n_samples * F, iph = n / halves;
  const int dt = net->cfg.dtype;
  RC(run_conv(net, net->layers[net->l_conv1], packed, t_in, acts + A.r_c1, n, nullptr, 0, nullptr, st));
  RC(run_bn(net, net->l_conv1, params, bnstats, acts, A, acts + A.r_c1, nullptr, acts + A.c1, n, iph, 1, st));
  RC(vdqn_maxpool_fwd(acts + A.c1, acts + A.pool, acts + A.idx, n, 112, 112, 64, dt, st));
  const unsigned char* x = acts + A.pool;
  for (int b = 0; b < 8; ++b) {
    const int i1 = net->l_b_conv1[b], i2 = net->l_b_conv2[b], ids = net->l_b_ds[b];
    RC(run_conv(net, net->layers[i1], packed, x, acts + A.r_h[b], n, nullptr, 0, nullptr, st));
    RC(run_bn(net, i1, params, bnstats, acts, A, acts + A.r_h[b], nullptr, acts + A.h[b], n, iph, 1, st));
    RC(run_conv(net, net->layers[i2], packed, acts + A.h[b], acts + A.r_o[b], n, nullptr, 0, nullptr, st));
    const void* identity = x;
    if (ids >= 0) {
      RC(run_conv(net, net->layers[ids], packed, x, acts + A.r_ds[b], n, nullptr, 0, nullptr, st));
      RC(run_bn(net, ids, params, bnstats, acts, A, acts + A.r_ds[b], nullptr, acts + A.ds[b], n, iph, 0, st));
      identity = acts + A.ds[b];
    }
    RC(run_bn(net, i2, params, bnstats, acts, A, acts + A.r_o[b], identity, acts + A.o[b], n, iph, 1, st));
    x = acts + A.o[b];
  }
  RC(vdqn_avgpool_fwd(x, acts + A.avg, n, 49, 512, dt, st));
  RC(run_conv(net, net->layers[net->l_top4], packed, acts + A.avg, acts + A.q, n_samples, nullptr, 0, reinterpret_cast<float*>(acts + A.qf), st));
  return VDQN_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" int vdqn_net_create(const vdqn_net_config* cfg, vdqn_net** out) {
  VDQN_CHECK(cfg && out, "vdqn_net_create: null arg");
  VDQN_CHECK(cfg->extra_capacity == 0 || cfg->extra_capacity == 1, "vdqn_net_create: extra_capacity must be 0 or 1");
  VDQN_CHECK(cfg->dtype == VDQN_F32 || cfg->dtype == VDQN_BF16, "vdqn_net_create: bad dtype %d", cfg->dtype);
  VDQN_CHECK(cfg->action_dim >= 1 && cfg->num_classes >= 1 && cfg->action_dim * cfg->num_classes <= 64, "vdqn_net_create: action_dim*num_classes must be in 1..64");
  VDQN_CHECK(cfg->num_frames >= 1 && cfg->num_frames <= 64, "vdqn_net_create: num_frames out of range");
  VDQN_CHECK(cfg->max_batch >= 1, "vdqn_net_create: max_batch");
  VDQN_CHECK(cfg->deterministic == 0 || cfg->deterministic == 1, "vdqn_net_create: deterministic must be 0 or 1");
  vdqn_net* net = new vdqn_net();
  net->cfg = *cfg;
  net->esz = cfg->dtype == VDQN_BF16 ? 2 : 4;
  {
    const char* no = getenv("VDQN_NO_OVERLAP");
    net->overlap = (no && no[0] == '1') ? 0 : 1;
  }
  build_layers(net);
  if ((int)net->layers.size() > kMaxLayers) {
    delete net;
    vdqn_set_error("vdqn_net_create: layer table overflow");
    return VDQN_ERR_INVALID;
  }
  *out = net;
  return VDQN_OK;
}

extern "C" int vdqn_net_set_overlap(vdqn_net* net, int on) {
  VDQN_CHECK(net, "vdqn_net_set_overlap: null net");
  if (net->side) (void)hipStreamSynchronize(net->side);
  if (net->side2) (void)hipStreamSynchronize(net->side2);
  net->overlap = on ? 1 : 0;
  return VDQN_OK;
}

extern "C" int vdqn_net_set_bn_sync(vdqn_net* net, vdqn_allreduce_fn fn, void* user, int32_t world_size) {
  VDQN_CHECK(net, "vdqn_net_set_bn_sync: null net");
  VDQN_CHECK(net->basic() || fn == nullptr, "vdqn_net_set_bn_sync: only ARCHITECTURE='basic' has train-mode BatchNorm");
  net->bn_sync.fn = (fn && world_size > 1) ? fn : nullptr;
  net->bn_sync.user = user;
  net->bn_sync.world = world_size > 1 ? world_size : 1;
  return VDQN_OK;
}

extern "C" void vdqn_net_destroy(vdqn_net* net) {
  if (!net) return;
  if (net->side) {
    (void)hipStreamSynchronize(net->side);
    for (auto& e : net->events)
      if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(net->side);
    if (net->side2) {
      (void)hipStreamSynchronize(net->side2);
      (void)hipStreamDestroy(net->side2);
    }
  }
  delete net;
}

extern "C" int vdqn_net_num_params(const vdqn_net* net) { return net ? (int)net->params.size() : 0; }
extern "C" int vdqn_net_param_info(const vdqn_net* net, int index, vdqn_param_info* out) {
  VDQN_CHECK(net && out && index >= 0 && index < (int)net->params.size(), "vdqn_net_param_info: bad index");
  *out = net->params[index];
  return VDQN_OK;
}
extern "C" int64_t vdqn_net_params_numel(const vdqn_net* net) { return net->params_numel; }
extern "C" int64_t vdqn_net_trainable_numel(const vdqn_net* net) { return net->trainable_numel; }
extern "C" int64_t vdqn_net_bnstats_numel(const vdqn_net* net) { return net->bnstats_numel; }
extern "C" int vdqn_net_stage_range(const vdqn_net* net, int stage, int64_t* begin, int64_t* end) {
  VDQN_CHECK(net && stage >= 0 && stage < 3 && begin && end, "vdqn_net_stage_range: bad args");
  *begin = net->stage_begin[stage];
  *end = net->stage_end[stage];
  return VDQN_OK;
}
extern "C" int64_t vdqn_net_packed_bytes(const vdqn_net* net) { return net->packed_bytes; }
extern "C" int64_t vdqn_net_acts_bytes(const vdqn_net* net, int32_t n_samples) { return act_layout(net, n_samples).total; }
extern "C" int64_t vdqn_net_bwd_bytes(const vdqn_net* net, int32_t n_samples) { return bwd_layout(net, n_samples).total; }

static int64_t indexed(const char* name, const char* prefix, const int64_t* arr) {
  const size_t n = strlen(prefix);
  if (strncmp(name, prefix, n) != 0 || name[n] < '0' || name[n] > '7' || name[n + 1] != 0) return -2;
  return arr[name[n] - '0'];
}
extern "C" int64_t vdqn_net_act_offset(const vdqn_net* net, int32_t n_samples, const char* name) {
  if (!net || !name) return -1;
  const ActLayout A = act_layout(net, n_samples);
  const struct { const char* n; int64_t v; } tab[] = {{"t_in", A.t_in}, {"c1", A.c1}, {"pool", A.pool}, {"idx", A.idx}, {"f8", A.f8},
                                                     {"l0", A.l0}, {"l1", A.l1}, {"q", A.q}, {"qf", A.qf}};
  for (auto& t : tab)
    if (strcmp(t.n, name) == 0) return t.v;
  int64_t v;
  if ((v = indexed(name, "h", A.h)) != -2) return v;
  if ((v = indexed(name, "o", A.o)) != -2) return v;
  if ((v = indexed(name, "ds", A.ds)) != -2) return v;
  if (strcmp(name, "avg") == 0) return A.avg;
  if (strcmp(name, "r_c1") == 0) return A.r_c1;
  if ((v = indexed(name, "r_h", A.r_h)) != -2) return v;
  if ((v = indexed(name, "r_o", A.r_o)) != -2) return v;
  if ((v = indexed(name, "r_ds", A.r_ds)) != -2) return v;
  return -1;
}
extern "C" int64_t vdqn_net_bwd_offset(const vdqn_net* net, int32_t n_samples, const char* name) {
  if (!net || !name) return -1;
  const BwdLayout W = bwd_layout(net, n_samples);
  const struct { const char* n; int64_t v; } tab[] = {{"dq", W.dq}, {"g_l1", W.g_l1}, {"g_l0", W.g_l0}, {"g_f8", W.g_f8},
                                                     {"g_pool", W.g_pool}, {"g_c1", W.g_c1}};
  for (auto& t : tab)
    if (strcmp(t.n, name) == 0) return t.v;
  int64_t v;
  if ((v = indexed(name, "g_o", W.g_o)) != -2) return v;
  if ((v = indexed(name, "g_h", W.g_h)) != -2) return v;
  if ((v = indexed(name, "dsg", W.dsg)) != -2) return v;
  if (strcmp(name, "g_avg") == 0) return W.g_avg;
  if ((v = indexed(name, "g_or", W.g_or)) != -2) return v;
  if ((v = indexed(name, "g_dsr", W.g_dsr)) != -2) return v;
  if (strncmp(name, "dw:", 3) == 0 || strncmp(name, "db:", 3) == 0)
    for (auto& L : net->layers)
      if (L.name == name + 3) return name[1] == 'w' ? L.dw_off : L.db_off;
  return -1;
}

// BatchNorm fold + layout packs of layers [first_layer, first_layer + n_layers) of the table (stored by backward stage: head +
// layer4, layer3, then stem + layer1 + layer2)
static int pack_weights_layers(vdqn_net* net, const float* params, const float* bnstats, void* packed, int32_t with_dgrad, int first_layer,
                               int n_layers, hipStream_t stream) {
  if (n_layers <= 0) return VDQN_OK;
  dim3 grid(256, (unsigned)n_layers, 2);
  const int dgrad = with_dgrad & 1, raw = (with_dgrad >> 1) & 1;
  int tile_first = -1, tile_end = 0;
  for (int i = first_layer; i < first_layer + n_layers; ++i) {
    const FoldDesc& d = net->fold.d[i];
    if (!d.tiled) continue;
    if (tile_first < 0) tile_first = d.tile_begin;
    tile_end = d.tile_begin + (d.co_pad / 64) * (d.ci / 64);
  }
  const double share = (double)n_layers / (double)net->layers.size();
  ProfScope ps_("fold_weights", 0.0, ((double)net->trainable_numel * 4.0 + (double)net->packed_bytes * (dgrad ? 1.0 : 0.5)) * share, stream);
  const size_t tile_smem = kFoldCot * (64 * 9 + 1) * 4;  // [kFoldCot output channels][64 * taps + 1] f32
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&fold_tile_kernel<bf16raw>), (size_t)tile_smem);
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&fold_tile_kernel<float>), (size_t)tile_smem);
  if (net->cfg.dtype == VDQN_BF16) {
    hipLaunchKernelGGL((fold_kernel<bf16raw>), grid, dim3(256), 0, stream, net->fold, params, bnstats, (unsigned char*)packed, dgrad, raw, first_layer);
    if (tile_first >= 0)
      hipLaunchKernelGGL((fold_tile_kernel<bf16raw>), dim3(tile_end - tile_first, 64 / kFoldCot), dim3(256), tile_smem, stream, net->fold, params, bnstats,
                         (unsigned char*)packed, dgrad, raw, tile_first);
  } else {
    hipLaunchKernelGGL((fold_kernel<float>), grid, dim3(256), 0, stream, net->fold, params, bnstats, (unsigned char*)packed, dgrad, raw, first_layer);
    if (tile_first >= 0)
      hipLaunchKernelGGL((fold_tile_kernel<float>), dim3(tile_end - tile_first, 64 / kFoldCot), dim3(256), tile_smem, stream, net->fold, params, bnstats,
                         (unsigned char*)packed, dgrad, raw, tile_first);
  }
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_net_pack_weights(vdqn_net* net, const float* params, const float* bnstats, void* packed, int32_t with_dgrad, void* stream) {
  VDQN_CHECK(net && params && bnstats && packed, "vdqn_net_pack_weights: null arg");
  return pack_weights_layers(net, params, bnstats, packed, with_dgrad, 0, (int)net->layers.size(), (hipStream_t)stream);
}

extern "C" int vdqn_net_forward(vdqn_net* net, const void* packed, const void* frames, int32_t src_kind, int32_t n_samples, void* acts,
                                float* q_out, void* stream) {
  VDQN_CHECK(net && packed && frames && acts && q_out, "vdqn_net_forward: null arg");
  VDQN_CHECK(n_samples >= 1 && n_samples <= net->cfg.max_batch, "vdqn_net_forward: n_samples %d exceeds max_batch %d", n_samples, net->cfg.max_batch);
  hipStream_t st = (hipStream_t)stream;
  const ActLayout A = act_layout(net, n_samples);
  unsigned char* ab = (unsigned char*)acts;
  RC(vdqn_pack_input(frames, src_kind, ab + A.t_in, n_samples * net->cfg.num_frames, net->cfg.dtype, st));
  RC(forward_impl(net, (const unsigned char*)packed, ab + A.t_in, n_samples, ab, A, st));
  const int nq = net->cfg.action_dim * net->cfg.num_classes;
  hipError_t e = hipMemcpy2DAsync(q_out, (size_t)nq * 4, ab + A.qf, 64 * 4, (size_t)nq * 4, (size_t)n_samples, hipMemcpyDeviceToDevice, st);
  VDQN_CHECK(e == hipSuccess, "vdqn_net_forward: q copy failed: %s", hipGetErrorString(e));
  return VDQN_OK;
}

extern "C" int vdqn_net_trunk_forward(vdqn_net* net, const void* packed, const void* frames, int32_t src_kind, int32_t n_samples, void* acts,
                                      void* stream) {
  VDQN_CHECK(net && packed && frames && acts, "vdqn_net_trunk_forward: null arg");
  VDQN_CHECK(n_samples >= 1 && n_samples <= net->cfg.max_batch, "vdqn_net_trunk_forward: n_samples %d exceeds max_batch %d", n_samples, net->cfg.max_batch);
  hipStream_t st = (hipStream_t)stream;
  const ActLayout A = act_layout(net, n_samples);
  unsigned char* ab = (unsigned char*)acts;
  RC(vdqn_pack_input(frames, src_kind, ab + A.t_in, n_samples * net->cfg.num_frames, net->cfg.dtype, st));
  return forward_impl(net, (const unsigned char*)packed, ab + A.t_in, n_samples, ab, A, st, true);
}

extern "C" int vdqn_net_forward_train(vdqn_net* net, const float* params, float* bnstats, void* packed, const void* frames, int32_t src_kind,
                                      int32_t n_samples, void* acts, float* q_out, void* stream) {
  VDQN_CHECK(net && params && bnstats && packed && frames && acts && q_out, "vdqn_net_forward_train: null arg");
  VDQN_CHECK(net->basic(), "vdqn_net_forward_train: only ARCHITECTURE='basic' has train-mode BatchNorm (extra_capacity: use vdqn_net_forward)");
  VDQN_CHECK(n_samples >= 1 && n_samples <= net->cfg.max_batch, "vdqn_net_forward_train: n_samples %d exceeds max_batch %d", n_samples, net->cfg.max_batch);
  hipStream_t st = (hipStream_t)stream;
  const ActLayout A = act_layout(net, n_samples);
  unsigned char* ab = (unsigned char*)acts;
  RC(vdqn_net_pack_weights(net, params, bnstats, packed, 2, st));
  RC(vdqn_pack_input(frames, src_kind, ab + A.t_in, n_samples * net->cfg.num_frames, net->cfg.dtype, st));
  RC(forward_train_impl(net, (const unsigned char*)packed, params, bnstats, ab + A.t_in, n_samples, 1, ab, A, st));
  const int nq = net->cfg.action_dim * net->cfg.num_classes;
  hipError_t e = hipMemcpy2DAsync(q_out, (size_t)nq * 4, ab + A.qf, 64 * 4, (size_t)nq * 4, (size_t)n_samples, hipMemcpyDeviceToDevice, st);
  VDQN_CHECK(e == hipSuccess, "vdqn_net_forward_train: q copy failed: %s", hipGetErrorString(e));
  return VDQN_OK;
}

// samples the `acts_online` workspace of this update is laid out for
static int step_layout_samples(const vdqn_net* net, const vdqn_step_args* a) {
  if (a->acts_samples > 0) return a->acts_samples;  // the workspace of one vdqn_net_forward call (vdqn_net_backward_begin)
  if (a->train_on_ground_truth) return a->batch;
  return 2 * a->batch;
}

extern "C" int vdqn_net_td_forward(vdqn_net* net, const vdqn_step_args* a, void* stream) {
  VDQN_CHECK(net && a, "vdqn_net_td_forward: null arg");
  VDQN_CHECK(a->params && a->bnstats && a->packed_online && a->before && a->act && a->acts_online && a->bwd && a->loss, "vdqn_net_td_forward: null buffer");
  const int B = a->batch;
  const bool gtb = a->train_on_ground_truth != 0;
  VDQN_CHECK(B >= 1 && 2 * B <= net->cfg.max_batch, "vdqn_net_td_forward: batch %d needs max_batch >= %d", B, 2 * B);
  VDQN_CHECK(gtb ? (a->gt != nullptr) : (a->after && a->packed_target && a->rew && a->term), "vdqn_net_td_forward: missing inputs for this loss branch");
  VDQN_CHECK(gtb || a->acts_target, "vdqn_net_td_forward: acts_target is NULL");
  hipStream_t st = (hipStream_t)stream;
  const int F = net->cfg.num_frames, dt = net->cfg.dtype;
  const int ns_online = step_layout_samples(net, a);
  const ActLayout A = act_layout(net, ns_online);
  const BwdLayout W = bwd_layout(net, B);
  net->bwd_samples = B;
  unsigned char* ao = (unsigned char*)a->acts_online;
  unsigned char* bw = (unsigned char*)a->bwd;

  // the frames are packed on the side stream while the main stream folds the weights (two small kernels each); the target
  // pass then simply continues on the side stream: it only reads the packed input and its own weights
  const int64_t frame_bytes = (int64_t)115 * 115 * 16 * net->esz;
  hipStream_t tst = fork_side(net, st);  // == st when the overlap is off
  // the packed frames of this update: the engine's own buffer, or the caller's (vdqn_step_args.packed_frames: packed ahead of time)
  const unsigned char* tin = a->packed_frames ? (const unsigned char*)a->packed_frames : ao + A.t_in;
  // (tried and measured slower, experiments/: the s' frames packed first with the target pass right behind them; the two packs
  // on two streams; a split weight fold with layer3+ beside the stem; stage folds behind their early Adam; online and target
  // forward as one chain of grouped launches; the online pass as two half-batch passes on two streams)
  if (!a->packed_frames) {
    RC(vdqn_pack_input(a->before, a->src_kind, ao + A.t_in, B * F, dt, tst));
    if (!gtb) RC(vdqn_pack_input(a->after, a->src_kind, ao + A.t_in + (int64_t)B * F * frame_bytes, B * F, dt, tst));
  }
  RC(vdqn_net_pack_weights(net, a->params, a->bnstats, a->packed_online, net->basic() ? 3 : 1, st));
  if (tst != st) join_side(net, st);  // packed input ready for the online pass
  if (!gtb) {
    const ActLayout T = act_layout(net, B);
    RC(forward_impl(net, (const unsigned char*)a->packed_target, tin + (int64_t)B * F * frame_bytes, B, (unsigned char*)a->acts_target, T, tst, false, 0));  // (no backward: no arg-max bytes)
  }
  if (net->basic())  // two model calls (before, after), each with its own batch statistics; running stats updated in place
    RC(forward_train_impl(net, (const unsigned char*)a->packed_online, a->params, a->bnstats, tin, ns_online, gtb ? 1 : 2, ao, A, st));
  else
    RC(forward_impl(net, (const unsigned char*)a->packed_online, tin, ns_online, ao, A, st, false, B));
  if (tst != st) join_side(net, st);

  // (clearing the 47 MB of accumulators on a side stream beside the packs instead of here, between the forward pass and the loss,
  // measured no gain: profiles/r6_12_ab_inproc_fused_head_and_clear_placement.txt)
  hipError_t e = hipMemsetAsync(bw + W.zero_begin, 0, (size_t)W.zero_bytes, st);
  VDQN_CHECK(e == hipSuccess, "vdqn_net_td_forward: memset failed: %s", hipGetErrorString(e));
  e = hipMemsetAsync(a->loss, 0, 4, st);
  VDQN_CHECK(e == hipSuccess, "vdqn_net_td_forward: memset failed: %s", hipGetErrorString(e));

  const float* qf_online = reinterpret_cast<const float*>(ao + A.qf);
  if (!gtb) {
    const ActLayout T = act_layout(net, B);
    unsigned char* at = (unsigned char*)a->acts_target;
    vdqn_td_args t;
    memset(&t, 0, sizeof(t));
    t.q_before = qf_online;
    t.q_after_online = qf_online + (size_t)B * 64;
    t.q_after_target = reinterpret_cast<const float*>(at + T.qf);
    t.act = a->act; t.rew = a->rew; t.term = a->term; t.valid = a->valid;
    t.loss = a->loss;
    t.dq = bw + W.dq;
    t.batch = B; t.n_cat = net->cfg.num_classes; t.n_act = net->cfg.action_dim; t.ldq = 64;
    t.gamma = a->gamma; t.inv_count = a->inv_count;
    t.clip_rect = a->clip_rect; t.linear = a->linear; t.use_valid = a->use_valid; t.dtype = dt;
    t.loss_kind = a->loss_kind;
    t.deterministic = net->cfg.deterministic;
    t.q_copy = a->q_before;  // (the compact copy of Q(s) rides in the loss launch: no 2-D copy between the loss and the first data gradient)
    RC(vdqn_td_loss(&t, st));
  } else {
    RC(vdqn_gt_loss(qf_online, a->act, a->gt, a->loss, bw + W.dq, nullptr, B, net->cfg.num_classes, net->cfg.action_dim, 64, a->inv_count,
                    a->value_learning, dt, st));
  }
  if (a->q_before && gtb) {
    const int nq = net->cfg.action_dim * net->cfg.num_classes;
    e = hipMemcpy2DAsync(a->q_before, (size_t)nq * 4, qf_online, 64 * 4, (size_t)nq * 4, (size_t)B, hipMemcpyDeviceToDevice, st);
    VDQN_CHECK(e == hipSuccess, "vdqn_net_td_forward: q copy failed: %s", hipGetErrorString(e));
  }
  return VDQN_OK;
}

namespace {

// backward of BasicBlock b (gradient of its output, already ReLU-masked, is in g_o[b])
int block_backward(vdqn_net* net, const vdqn_step_args* a, int b, const ActLayout& A, const BwdLayout& W, int n, hipStream_t st) {
  const unsigned char* pk = (const unsigned char*)a->packed_online;
  unsigned char* ao = (unsigned char*)a->acts_online;
  unsigned char* bw = (unsigned char*)a->bwd;
  const Layer& c1 = net->layers[net->l_b_conv1[b]];
  const Layer& c2 = net->layers[net->l_b_conv2[b]];
  const unsigned char* x = b == 0 ? ao + A.pool : ao + A.o[b - 1];
  unsigned char* gx = b == 0 ? bw + W.g_pool : bw + W.g_o[b - 1];
  const void* g_out = bw + W.g_o[b];
  // conv2: weight gradient, then data gradient into g_h masked by relu(h)
  hipStream_t ws = wgrad_stream(net, st);  // g_out is complete on `st`
  RC(run_wgrad(net, c2, bw, g_out, ao + A.h[b], n, ws));
  if (net->l_b_ds[b] >= 0) RC(run_wgrad(net, net->layers[net->l_b_ds[b]], bw, g_out, x, n, ws));
  RC(run_dgrad(net, c2, pk, g_out, bw + W.g_h[b], n, nullptr, ao + A.h[b], st, bw + W.p_h[b]));
  ws = wgrad_stream(net, st);  // g_h is complete
  RC(run_wgrad(net, c1, bw, bw + W.g_h[b], x, n, ws));
  const void* resid = g_out;  // identity shortcut
  if (net->l_b_ds[b] >= 0) {
    const Layer& ds = net->layers[net->l_b_ds[b]];
    if (fuse_ds()) {  // the shortcut's gradient is the downsample's data gradient: summed inside conv1's data-gradient launch
      RC(run_dgrad(net, c1, pk, bw + W.g_h[b], gx, n, nullptr, x, st, b > 0 ? bw + W.p_o[b - 1] : bw + W.p_pool, &ds, g_out));
      return VDQN_OK;
    }
    RC(run_dgrad(net, ds, pk, g_out, bw + W.dsg[b], n, nullptr, nullptr, st));
    resid = bw + W.dsg[b];
  }
  RC(run_dgrad(net, c1, pk, bw + W.g_h[b], gx, n, resid, x, st, b > 0 ? bw + W.p_o[b - 1] : bw + W.p_pool));
  return VDQN_OK;
}

// train-mode BatchNorm backward of layer li on the `before` half: dy = d/dy of bn(y), parameter gradients straight into `grads`
int run_bn_bwd(const vdqn_net* net, const vdqn_step_args* a, int li, const ActLayout& A, const void* g, const void* y, void* dy, int n,
               hipStream_t st) {
  const Layer& L = net->layers[li];
  BnSync sy = net->bn_sync;
  sy.scratch = reinterpret_cast<float*>((unsigned char*)a->acts_online + A.bn_sync);
  if (A.bn_det >= 0) {
    sy.det_ws = reinterpret_cast<float*>((unsigned char*)a->acts_online + A.bn_det);
    sy.det_ws_bytes = A.bn_det_bytes;
  }
  return vdqn_bn_train_bwd_impl(g, y, dy, reinterpret_cast<float*>((unsigned char*)a->acts_online + A.bnw[li]), a->grads + L.g_off, a->grads + L.b_off,
                                n, L.ho * L.wo, L.co, net->cfg.num_frames, n, net->cfg.dtype, st, &sy);
}

// 'basic': BasicBlock b with train-mode BatchNorm; g_o[b] holds the (ReLU-masked) gradient of the block output
int block_backward_train(vdqn_net* net, const vdqn_step_args* a, int b, const ActLayout& A, const BwdLayout& W, int n, hipStream_t st) {
  const unsigned char* pk = (const unsigned char*)a->packed_online;
  unsigned char* ao = (unsigned char*)a->acts_online;
  unsigned char* bw = (unsigned char*)a->bwd;
  const int i1 = net->l_b_conv1[b], i2 = net->l_b_conv2[b], ids = net->l_b_ds[b];
  const Layer& c1 = net->layers[i1];
  const Layer& c2 = net->layers[i2];
  const unsigned char* x = b == 0 ? ao + A.pool : ao + A.o[b - 1];
  unsigned char* gx = b == 0 ? bw + W.g_pool : bw + W.g_o[b - 1];
  const void* g_out = bw + W.g_o[b];
  RC(run_bn_bwd(net, a, i2, A, g_out, ao + A.r_o[b], bw + W.g_or[b], n, st));
  hipStream_t ws = fork_side(net, st);
  RC(run_wgrad(net, c2, bw, bw + W.g_or[b], ao + A.h[b], n, ws));
  if (ids >= 0) {
    RC(run_bn_bwd(net, a, ids, A, g_out, ao + A.r_ds[b], bw + W.g_dsr[b], n, st));
    ws = fork_side(net, st);
    RC(run_wgrad(net, net->layers[ids], bw, bw + W.g_dsr[b], x, n, ws));
  }
  RC(run_dgrad(net, c2, pk, bw + W.g_or[b], bw + W.g_h[b], n, nullptr, ao + A.h[b], st));
  RC(run_bn_bwd(net, a, i1, A, bw + W.g_h[b], ao + A.r_h[b], bw + W.g_h[b], n, st));
  ws = fork_side(net, st);
  RC(run_wgrad(net, c1, bw, bw + W.g_h[b], x, n, ws));
  const void* resid = g_out;
  if (ids >= 0) {
    RC(run_dgrad(net, net->layers[ids], pk, bw + W.g_dsr[b], bw + W.dsg[b], n, nullptr, nullptr, st));
    resid = bw + W.dsg[b];
  }
  RC(run_dgrad(net, c1, pk, bw + W.g_h[b], gx, n, resid, x, st));
  return VDQN_OK;
}

}  // namespace

// dL/dQ f32 [B][nq] -> the engine's [B][64] operand of the head's backward (zero padded)
template <typename T>
__global__ __launch_bounds__(256) void dq_pad_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int nq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 64) return;
  const int b = i >> 6, c = i & 63;
  dst[i] = from_f32<T>(c < nq ? src[(long)b * nq + c] : 0.f);
}

extern "C" int vdqn_net_backward_begin(vdqn_net* net, const vdqn_step_args* a, const float* dq_f32, void* stream) {
  VDQN_CHECK(net && a && dq_f32, "vdqn_net_backward_begin: null arg");
  VDQN_CHECK(!net->basic(), "vdqn_net_backward_begin: only extra_capacity (eval-mode BatchNorm) has a per-call backward; ARCHITECTURE='basic' trains through vdqn_net_td_forward");
  VDQN_CHECK(a->params && a->bnstats && a->packed_online && a->acts_online && a->bwd && a->grads, "vdqn_net_backward_begin: null buffer");
  VDQN_CHECK(a->batch >= 1 && a->batch <= net->cfg.max_batch && a->acts_samples == a->batch,
             "vdqn_net_backward_begin: acts_samples (%d) must equal batch (%d) <= max_batch", a->acts_samples, a->batch);
  hipStream_t st = (hipStream_t)stream;
  const int B = a->batch;
  const BwdLayout W = bwd_layout(net, B);
  net->bwd_samples = B;
  unsigned char* bw = (unsigned char*)a->bwd;
  hipError_t e = hipMemsetAsync(bw + W.zero_begin, 0, (size_t)W.zero_bytes, st);
  VDQN_CHECK(e == hipSuccess, "vdqn_net_backward_begin: memset failed: %s", hipGetErrorString(e));
  const int nq = net->cfg.action_dim * net->cfg.num_classes;
  const int blocks = (B * 64 + 255) / 256;
  if (net->cfg.dtype == VDQN_BF16)
    hipLaunchKernelGGL((dq_pad_kernel<bf16raw>), dim3(blocks), dim3(256), 0, st, dq_f32, reinterpret_cast<bf16raw*>(bw + W.dq), B, nq);
  else
    hipLaunchKernelGGL((dq_pad_kernel<float>), dim3(blocks), dim3(256), 0, st, dq_f32, reinterpret_cast<float*>(bw + W.dq), B, nq);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" void* vdqn_net_grad_stream(vdqn_net* net) {
  if (!net || !side_ready(net)) return nullptr;
  return (void*)net->side;
}

extern "C" int vdqn_net_backward_stage(vdqn_net* net, const vdqn_step_args* a, int32_t stage, void* stream) {
  VDQN_CHECK(net && a && a->grads, "vdqn_net_backward_stage: null arg");
  VDQN_CHECK(stage >= 0 && stage < 3, "vdqn_net_backward_stage: stage %d", stage);
  hipStream_t st = (hipStream_t)stream;
  const int B = a->batch, F = net->cfg.num_frames, n = B * F, dt = net->cfg.dtype;
  const bool gtb = a->train_on_ground_truth != 0;
  const ActLayout A = act_layout(net, step_layout_samples(net, a));
  const BwdLayout W = bwd_layout(net, B);
  const unsigned char* pk = (const unsigned char*)a->packed_online;
  unsigned char* ao = (unsigned char*)a->acts_online;
  unsigned char* bw = (unsigned char*)a->bwd;
  bool split_conv1 = false;  // stage 2, extra_capacity: conv1's weight gradient is unfolded separately (see below)
  int pr_l1 = 128, pr_l0 = 128, pr_f8 = 128;  // row granularity of the head's column-sum partials (stage 0)

  if (net->basic()) {
    if (stage == 0) {
      const Layer& top = net->layers[net->l_top4];
      RC(run_wgrad(net, top, bw, bw + W.dq, ao + A.avg, B, wgrad_stream(net, st), true));
      RC(run_dgrad(net, top, pk, bw + W.dq, bw + W.g_avg, B, nullptr, nullptr, st));
      RC(vdqn_avgpool_bwd(bw + W.g_avg, ao + A.o[7], bw + W.g_o[7], n, 49, 512, dt, st));
      RC(block_backward_train(net, a, 7, A, W, n, st));
      RC(block_backward_train(net, a, 6, A, W, n, st));
    } else if (stage == 1) {
      RC(block_backward_train(net, a, 5, A, W, n, st));
      RC(block_backward_train(net, a, 4, A, W, n, st));
    } else {
      for (int b = 3; b >= 0; --b) RC(block_backward_train(net, a, b, A, W, n, st));
      RC(vdqn_maxpool_bwd(bw + W.g_pool, ao + A.idx, nullptr, bw + W.g_c1, n, 112, 112, 64, dt, st));
      RC(run_bn_bwd(net, a, net->l_conv1, A, bw + W.g_c1, ao + A.r_c1, bw + W.g_c1, n, st));
      RC(run_wgrad(net, net->layers[net->l_conv1], bw, bw + W.g_c1, a->packed_frames ? a->packed_frames : ao + A.t_in, n, wgrad_stream(net, st)));
    }
  } else if (stage == 0) {
    const Layer& t4 = net->layers[net->l_top4];
    const Layer& t2 = net->layers[net->l_top2];
    const Layer& t0 = net->layers[net->l_top0];
    const Layer& f8 = net->layers[net->l_f8];
    RC(run_wgrad(net, t4, bw, bw + W.dq, ao + A.l1, B, wgrad_stream(net, st), true));
    RC(run_dgrad(net, t4, pk, bw + W.dq, bw + W.g_l1, B, nullptr, ao + A.l1, st, bw + W.p_l1, nullptr, nullptr, &pr_l1));
    RC(run_wgrad(net, t2, bw, bw + W.g_l1, ao + A.l0, B, wgrad_stream(net, st)));
    RC(run_dgrad(net, t2, pk, bw + W.g_l1, bw + W.g_l0, B, nullptr, ao + A.l0, st, bw + W.p_l0, nullptr, nullptr, &pr_l0));
    RC(run_wgrad(net, t0, bw, bw + W.g_l0, ao + A.f8, B, wgrad_stream(net, st)));
    RC(run_dgrad(net, t0, pk, bw + W.g_l0, bw + W.g_f8, B, nullptr, ao + A.f8, st, bw + W.p_f8, nullptr, nullptr, &pr_f8));
    RC(run_wgrad(net, f8, bw, bw + W.g_f8, ao + A.o[7], n, wgrad_stream(net, st)));
    RC(run_dgrad(net, f8, pk, bw + W.g_f8, bw + W.g_o[7], n, nullptr, ao + A.o[7], st, bw + W.p_o[7]));
    RC(block_backward(net, a, 7, A, W, n, st));
    RC(block_backward(net, a, 6, A, W, n, st));
  } else if (stage == 1) {
    RC(block_backward(net, a, 5, A, W, n, st));
    RC(block_backward(net, a, 4, A, W, n, st));
  } else {
    for (int b = 3; b >= 0; --b) RC(block_backward(net, a, b, A, W, n, st));
    // conv1's weight gradient is the last link of the chain (max-pool backward -> wgrad): the other layers of the stage are
    // unfolded ahead of it
    split_conv1 = net->overlap && net->side && net->l_conv1 == net->layer_stage_first[2] && net->layer_stage_count[2] > 1;
  }
  int max_co = 0;
  for (int i = net->layer_stage_first[stage]; i < net->layer_stage_first[stage] + net->layer_stage_count[stage]; ++i)
    max_co = net->layers[i].co > max_co ? net->layers[i].co : max_co;
  PartTable pt;
  memset(&pt, 0, sizeof(pt));
  if (!net->basic()) {
    auto tiles = [](int64_t rows) { return (int)((rows + 127) / 128); };
    auto set = [&](int li, int64_t off, int64_t rows, int ld, int groups, int gstride) {
      if (li < 0) return;
      pt.off[li] = off; pt.tiles[li] = tiles(rows); pt.ld[li] = ld; pt.groups[li] = groups; pt.gstride[li] = gstride;
    };
    set(net->l_top2, W.p_l1, B, 256, 1, 0);
    set(net->l_top0, W.p_l0, B, 512, 1, 0);
    set(net->l_f8, W.p_f8, B, 1600 * F, 25 * F, 64);
    if (stage == 0) {  // entries per pr_* rows, as the kernels that ran above wrote them
      pt.tiles[net->l_top2] = (B + pr_l1 - 1) / pr_l1;
      pt.tiles[net->l_top0] = (B + pr_l0 - 1) / pr_l0;
      pt.tiles[net->l_f8] = (B + pr_f8 - 1) / pr_f8;
    }
    set(net->l_conv1, W.p_pool, (int64_t)n * 56 * 56, 64, 1, 0);
    for (int b = 0; b < 8; ++b) {
      const int planes = 64 << (b / 2), sp = 56 >> (b / 2);
      set(net->l_b_conv2[b], W.p_o[b], (int64_t)n * sp * sp, planes, 1, 0);
      set(net->l_b_ds[b], W.p_o[b], (int64_t)n * sp * sp, planes, 1, 0);
      if (b + 1 < 8 && net->l_b_ds[b + 1] >= 0) {  // g_o[b] comes from a stride-2 dgrad: 4 parity classes of tiles
        const int t4 = 4 * tiles((int64_t)n * sp * sp / 4);
        pt.tiles[net->l_b_conv2[b]] = t4;
        if (net->l_b_ds[b] >= 0) pt.tiles[net->l_b_ds[b]] = t4;
      }
      set(net->l_b_conv1[b], W.p_h[b], (int64_t)n * sp * sp, planes, 1, 0);
    }
  }
  const bool stem_tail = stage == 2 && !net->basic();
  auto conv1_chain = [&]() -> int {  // conv1's weight gradient from the pooled gradient g_pool
    // g_pool is already masked by (pool > 0) in block 0's dgrad epilogue and c1[argmax] == pool, so the ReLU mask of c1 is implied.
    // bn1's shift gradient = column sums of g_c1 = column sums of g_pool (max-pool routes every pooled gradient element
    // to exactly one input position), which block 0's dgrad epilogue already produced as partials.
    const Layer& L1 = net->layers[net->l_conv1];
    // bf16: ONE kernel on the side stream builds the max-pool backward tiles in LDS (VDQN_FUSE_POOL_BWD=0: the two launches below)
    static const bool fuse_pool = [] { const char* e = getenv("VDQN_FUSE_POOL_BWD"); return !(e && e[0] == '0'); }();
    const int64_t det_need = net->cfg.deterministic ? vdqn_stem_wgrad_pool_workspace_bytes(n) : 0;
    if (fuse_pool && dt == VDQN_BF16 && (int64_t)n * 115 * 115 * 32 < (1ll << 31) && det_need <= W.det_ws_bytes) {
      prof_layer(L1, n);
      // Which stream: the side stream still holds block 0's last two weight gradients when the data-gradient chain ends, so in
      // the default mode this kernel runs on the CALLER's stream beside them (the update's tail on two streams instead of one;
      // its unfold below waits for it).  Deterministic mode keeps it on the side stream: the partial copies share one workspace.
      static const bool on_main_env = [] { const char* e = getenv("VDQN_STEM_WGRAD_MAIN"); return !(e && e[0] == '0'); }();
      const bool on_main = g_stem_wgrad_main_override >= 0 ? g_stem_wgrad_main_override != 0 : on_main_env;
      const bool main_st = on_main && split_conv1 && !net->cfg.deterministic;
      return vdqn_stem_wgrad_pool(bw + W.g_pool, ao + A.idx, a->packed_frames ? a->packed_frames : ao + A.t_in, reinterpret_cast<float*>(bw + L1.dw_off), n,
                                  net->cfg.deterministic ? bw + W.det_ws : nullptr, W.det_ws_bytes, main_st ? st : fork_side(net, st));
    }
    // max-pool backward on the caller's stream, the weight gradient behind it on the side stream
    RC(vdqn_maxpool_bwd(bw + W.g_pool, ao + A.idx, nullptr, bw + W.g_c1, n, 112, 112, 64, dt, st));
    RC(run_wgrad(net, L1, bw, bw + W.g_c1, a->packed_frames ? a->packed_frames : ao + A.t_in, n, wgrad_stream(net, st)));
    return VDQN_OK;
  };
  if (stem_tail && !split_conv1) RC(conv1_chain());
  // The unfold runs BEHIND the stage's weight gradients on the side stream (which first waits for the caller's stream: the
  // column-sum partials come from the data-gradient epilogues there), so the caller's stream goes straight on to the next stage's
  // data gradients; only stage 2 joins the side stream back (before Adam).  vdqn_net_grad_stream() is where a stage's range of
  // `grads` is complete.
  hipStream_t us = fork_side(net, st);  // == st when the overlap is off
  join_wgrad_streams(net);
  const double unfold_bytes = (double)(net->stage_end[stage] - net->stage_begin[stage]) * 12.0;
  if (split_conv1) {
    {
      ProfScope ps_("unfold_grads", 0.0, unfold_bytes, us);
      hipLaunchKernelGGL(unfold_kernel, dim3(max_co, net->layer_stage_count[stage] - 1), dim3(256), 0, us, net->fold, pt, net->layer_stage_first[stage] + 1,
                         a->params, a->bnstats, (const unsigned char*)bw, a->grads, 0);
    }
    RC(conv1_chain());
    join_wgrad_streams(net);
    (void)fork_side(net, st);  // conv1's weight gradient may have run on the caller's stream
    ProfScope ps_("unfold_grads", 0.0, 0.0, net->side);
    hipLaunchKernelGGL(unfold_kernel, dim3(net->layers[net->l_conv1].co, 1), dim3(256), 0, net->side, net->fold, pt, net->l_conv1, a->params, a->bnstats,
                       (const unsigned char*)bw, a->grads, 0);
  } else {
    ProfScope ps_("unfold_grads", 0.0, unfold_bytes, us);
    hipLaunchKernelGGL(unfold_kernel, dim3(max_co, net->layer_stage_count[stage]), dim3(256), 0, us, net->fold, pt, net->layer_stage_first[stage],
                       a->params, a->bnstats, (const unsigned char*)bw, a->grads, net->basic() ? 1 : 0);
  }
  if (stage == 2) join_side(net, st);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
