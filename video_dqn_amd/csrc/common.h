// Shared definitions for the gfx950 kernels of the video-dqn Q-learning hot path.
// CDNA4 only (wave64, MFMA 16x16x32 bf16 / 16x16x4 f32, 160 KiB LDS); no CUDA-compat layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vdqn.h"

typedef uint16_t bf16raw;  // storage type of a bf16 element

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <typename T> struct TInfo;
template <> struct TInfo<float> {
  static constexpr int kDtype = VDQN_F32;
  static constexpr int kPer16B = 4;
};
template <> struct TInfo<bf16raw> {
  static constexpr int kDtype = VDQN_BF16;
  static constexpr int kPer16B = 8;
};

__device__ __forceinline__ float bf16_to_f32(bf16raw v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16raw f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16raw, b);
}
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16raw>(bf16raw v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16raw from_f32<bf16raw>(float v) { return f32_to_bf16(v); }

// Exact unsigned division p / d for p < 2^24, d < 2^16 via one 64-bit multiply (M = ceil(2^40 / d)).
struct FastDiv {
  uint64_t mul;
  uint32_t div;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.div = d;
  f.mul = ((1ull << 40) + d - 1) / d;
  return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t p, const FastDiv& f) { return (uint32_t)(((uint64_t)p * f.mul) >> 40); }

// bijective XCD-aware remap of a 1-D block id: blocks that share an XCD (id % 8) get a contiguous
// range of logical ids, so tiles that re-read the same operand rows hit that XCD's private L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
  const uint32_t q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

void vdqn_set_error(const char* fmt, ...);

// Per-device launch state (profile.hip).  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's copy
// of a kernel, and the CU count differs per device, so neither may be cached process-wide: both are keyed by hipGetDevice()
// and guarded by a mutex (engines on several GPUs, launches from several host threads).
void vdqn_ensure_dyn_smem(const void* kernel, size_t bytes);
int vdqn_num_cus();

// launch profiler hooks (profile.hip); no-ops unless vdqn_profile_enable(1)
void vdqn_prof_begin(const char* tag, double flops, double bytes, hipStream_t st);
void vdqn_prof_end(hipStream_t st);
extern thread_local double g_prof_alg_flops;  // engine sets the algorithmic FLOPs of the next launch (padding excluded)
extern thread_local const char* g_prof_suffix;  // engine: per-layer profile rows when VDQN_PROFILE_LAYERS=1
struct ProfScope {
  hipStream_t st;
  ProfScope(const char* tag, double flops, double bytes, hipStream_t s) : st(s) { vdqn_prof_begin(tag, flops, bytes, s); }
  ~ProfScope() { vdqn_prof_end(st); }
};
// SyncBN hook of the train-mode BatchNorm kernels (bn_train.hip; set through vdqn_net_set_bn_sync)
struct BnSync {
  vdqn_allreduce_fn fn;  // SUM all-reduce of `count` floats in place, ordered on `stream`
  void* user;
  float* scratch;        // device buffer for the packed sums (>= groups * 2 * channels floats)
  int world;
  float* det_ws = nullptr;       // deterministic mode: per-block partial sums (vdqn_bn_train_workspace_bytes), else f32 atomics
  int64_t det_ws_bytes = 0;
};
int vdqn_bn_train_fwd_impl(const void* y, const void* resid, void* z, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* work, int32_t n_img, int32_t hw, int32_t c, int32_t num_frames, int32_t imgs_per_half,
                           int32_t relu, float momentum, float eps, int32_t dtype, void* stream, const BnSync* sync);
int vdqn_bn_train_bwd_impl(const void* g, const void* y, void* dy, float* work, float* dgamma, float* dbeta, int32_t n_img, int32_t hw, int32_t c,
                           int32_t num_frames, int32_t imgs_per_half, int32_t dtype, void* stream, const BnSync* sync);

// skinny.hip: the Q-head's small GEMMs (bf16).  kind 0 = not taken, 1 = linear layer (32 x 32 tiles), 2 = valid convolution with 64
// output columns (features.8 forward, 64 x 64 tiles)
int vdqn_skinny_kind(const vdqn_conv_args* a);
int vdqn_skinny_part_rows(int conv);
int vdqn_launch_skinny(const void* igemm_params, int kind, hipStream_t stream);

#define VDQN_CHECK(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      vdqn_set_error(__VA_ARGS__);   \
      return VDQN_ERR_INVALID;       \
    }                                \
  } while (0)
#define VDQN_LAUNCH_CHECK()                                         \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) {                                         \
      vdqn_set_error("launch failed: %s", hipGetErrorString(e_));   \
      return VDQN_ERR_LAUNCH;                                       \
    }                                                               \
  } while (0)
