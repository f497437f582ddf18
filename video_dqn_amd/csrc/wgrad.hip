// Weight-gradient implicit GEMM for gfx950:
//
//   dw[n][r][s][c] += sum_m gy[m, n] * x[pix(m, r, s), c]        (m = output pixel, the reduction index)
//
// Both operands are stored pixel-major (channels contiguous), i.e. transposed w.r.t. what an MFMA A/B
// fragment wants (k = pixel contiguous per lane).  The tiles are therefore staged into LDS exactly as they
// lie in HBM ([32 pixels][BT channels], full-line 16-byte loads) and the transpose is done by the LDS read:
// ds_read_b64_tr_b16 for bf16 (4 pixels x 16 channels per 16-lane group, delivered channel-per-lane), plain
// ds_read_b32 for the f32 16x16x4 MFMA whose fragment is one element per lane.
//
// Grid: x = (co tile, tap, ci tile), y = split of the pixel range; partial tiles are summed with
// global_atomic_add_f32 into the pre-zeroed f32 dw (all lanes of a row write 64 contiguous bytes).
// LDS rows are XOR-swizzled on the 16-byte chunk index so the transposing reads of a half-wave (8 pixel rows
// x 32 B) fall on distinct banks.
//
// Replaces the convolution_backward / addmm weight-gradient kernels behind loss.backward()
// (train_q_network.py:226).
#include <type_traits>

#include "common.h"

namespace {

struct WgradParams {
  const void* gy;
  const void* x;
  float* dw;
  int n_img, hi, wi, ci, pix_stride, ho, wo, co, ldg, r, s, stride, pad;
  int M, kchunk, taps, ci_tiles, gy_bytes, x_bytes, splitk;
  FastDiv d_howo, d_wo;
  // deterministic mode: every block stores its partial tile into its split's private copy of dw (ws + split * ws_stride
  // floats, plain stores), and wgrad_reduce_kernel sums the copies into dw in split order; nullptr = f32 atomics into dw
  float* ws;
  long long ws_stride;
  void* stamps;  // -DVDQN_STAMP builds only (tools/stamp_wgrad.py): per-workgroup phase cycles of the window kernel's K loop
};

// partial-sum sink of a block: an atomic into dw, or a plain store into the block's split copy (uniform branch)
__device__ __forceinline__ void wg_emit(float* dst, bool det, float v) {
#ifdef VDQN_WGRAD_NO_EMIT  // timing-only diagnostic build (results INVALID): the kernels without their partial-sum sink = the bound of
  asm volatile("" ::"v"(v), "v"(dst));  // any cheaper reduction scheme (tools/bench_wgrad.py, profiles/r6_08_*)
  return;
#endif
  if (det) *dst = v;
  else atomicAdd(dst, v);
}

template <typename T, int BT> __device__ __forceinline__ int wg_swz(int row) {
  if constexpr (sizeof(T) == 2) {
    if constexpr (BT == 128) return (row & 7) << 1;
    else return ((row >> 1) & 3) << 1;
  } else {
    return (row & 1) << 2;
  }
}

// pixels per staged K-step / 32: sized so each barrier covers >= 16 MFMAs per wave within 64 KiB of LDS
template <typename T, int BT> constexpr int wg_ksub() {
  return sizeof(T) == 2 ? (BT == 64 ? 4 : 2) : (BT == 64 ? 2 : 1);
}

template <typename T, int BT>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  constexpr int E16 = 16 / (int)sizeof(T);
  constexpr int RB = BT * (int)sizeof(T);  // bytes per LDS row (one pixel)
  constexpr int CPR = RB / 16;             // 16-byte chunks per row
  constexpr int KSUB = wg_ksub<T, BT>();   // 32-pixel MFMA sub-steps per staged K-step
  constexpr int KP = 32 * KSUB;            // pixels staged per K-step (one barrier per KP pixels)
  constexpr int NL = (KP * CPR) / 256;     // 16-byte chunks per thread per operand per K-step
  // WSPLIT (bf16, 64x64 tile): instead of a 2x2 wave grid of 32x32 sub-tiles (4 MFMAs per 8 transposing reads: LDS
  // bandwidth bound), every wave owns the whole 64x64 tile for ONE of the four 32-pixel sub-steps of a staged K-step
  // (16 MFMAs per 16 reads) and the four partial tiles are summed through LDS before the atomics.
  constexpr bool WSPLIT = (sizeof(T) == 2 && BT == 64);
  constexpr int NFR = WSPLIT ? BT / 16 : BT / 32;  // 16-wide fragments per wave per dim
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                 // [2][KP*RB]
  unsigned char* sB = smem + 2 * KP * RB;   // [2][KP*RB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, XCD-aware: the blocks of one pixel range (all taps / channel tiles of one split) get consecutive
  // logical ids on ONE XCD, so the 9 taps re-read their gy / x rows from that XCD's L2 instead of the fabric.
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int n_tiles = (int)(gridDim.x / (uint32_t)p.splitk);
  const int split = (int)(lb / (uint32_t)n_tiles);
  int t = (int)(lb - (uint32_t)split * (uint32_t)n_tiles);
  const int ci_tile = t % p.ci_tiles;
  t /= p.ci_tiles;
  const int tap = t % p.taps;
  const int co_tile = t / p.taps;
  const int kr = tap / p.s, ks = tap - kr * p.s;
  const int co0 = co_tile * BT, ci0 = ci_tile * BT;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.M, kbeg + p.kchunk);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + KP - 1) / KP;

  const T* __restrict__ gy = (const T*)p.gy;
  const T* __restrict__ x = (const T*)p.x;

  // HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) issued from inline asm (see igemm.hip): the LDS image is
  // lane-linear ([32 pixels][RB bytes], thread q stages 16-byte slot q), so the XOR swizzle is applied to the
  // SOURCE chunk; rows past the split range and padding taps get an out-of-range offset -> zero fill.
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long g_ptr = (unsigned long long)p.gy, x_ptr = (unsigned long long)p.x;
  const i32x4 rs_g = {__builtin_amdgcn_readfirstlane((int)(unsigned)g_ptr), __builtin_amdgcn_readfirstlane((int)((g_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.gy_bytes), 0x00020000};
  const i32x4 rs_x = {__builtin_amdgcn_readfirstlane((int)(unsigned)x_ptr), __builtin_amdgcn_readfirstlane((int)((x_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.x_bytes), 0x00020000};
  constexpr unsigned kOobW = 0x80000000u;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t lds_wave = lds_base + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
  const uint32_t ldg_b = (uint32_t)p.ldg * (uint32_t)sizeof(T);
  const uint32_t pix_b = (uint32_t)p.pix_stride * (uint32_t)sizeof(T);

  // Staging state of this thread's NL 16-byte slots, carried from K-step to K-step (round 5).  Slot i stages pixel row
  // pm = kb + l_row[i]; its output position (img, oh, ow) and byte offsets used to be rebuilt from pm at every K-step by two
  // 64-bit-multiply divisions per slot — with 16 MFMAs per wave and K-step that was 11 vector instructions per MFMA, most of them
  // quarter-rate multiplies (MFMA-busy 7.6 %, profiles/r04bf_pmc_mfma.json).  K-steps are issued in order, KP pixels apart, so
  // the position advances by a FIXED (d_img, d_oh, d_ow) with at most one carry out of the column and one out of the row: adds,
  // compares and selects only; the divisions run once per workgroup.
  //   s_pm  pixel row (for the range test)         s_h, s_w  input row / column of tap (kr, ks): oh * stride - pad + kr, ...
  //   s_gb  byte offset of the gy chunk            s_xb      byte offset of the x chunk of that tap (valid or not)
  int s_pm[NL], s_h[NL], s_w[NL];
  uint32_t s_gb[NL], s_xb[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int q = tid + 256 * i;
    const int row = q / CPR;
    const int ch = (q % CPR) ^ wg_swz<T, BT>(row);
    const int pm = kbeg + row;
    const uint32_t img = fastdiv((uint32_t)pm, p.d_howo);
    const uint32_t rem = (uint32_t)pm - img * p.d_howo.div;
    const uint32_t oh = fastdiv(rem, p.d_wo);
    const uint32_t ow = rem - oh * p.d_wo.div;
    s_pm[i] = pm;
    s_h[i] = (int)oh * p.stride - p.pad + kr;
    s_w[i] = (int)ow * p.stride - p.pad + ks;
    s_gb[i] = (uint32_t)pm * ldg_b + (uint32_t)((co0 + ch * E16) * (int)sizeof(T));
    s_xb[i] = (uint32_t)(((int)(img * (uint32_t)p.hi) + s_h[i]) * p.wi + s_w[i]) * pix_b + (uint32_t)((ci0 + ch * E16) * (int)sizeof(T));
  }
  // advance of KP pixels (scalars; rows and columns in INPUT units, i.e. times the stride)
  const int adv_img = KP / (p.ho * p.wo), adv_rem = KP - adv_img * (p.ho * p.wo);
  const int adv_oh = adv_rem / p.wo, adv_ow = adv_rem - adv_oh * p.wo;
  const int adv_h = adv_oh * p.stride, adv_w = adv_ow * p.stride;
  const int wrap_w = p.wo * p.stride, wrap_h = p.ho * p.stride;  // what a wrapping column / row loses
  const int lim_w = wrap_w - p.pad + ks, lim_h = wrap_h - p.pad + kr;  // s_w >= lim_w <=> ow >= wo;  s_h >= lim_h <=> oh >= ho
  const uint32_t adv_x = (uint32_t)((adv_img * p.hi + adv_h) * p.wi + adv_w) * pix_b;
  const uint32_t car_ow = (uint32_t)((p.wi - p.wo) * p.stride) * pix_b;               // column wraps: ow -= wo, oh += 1
  const uint32_t car_oh = (uint32_t)((p.hi - p.ho * p.stride) * p.wi) * pix_b;        // row wraps:    oh -= ho, img += 1
  const uint32_t adv_g = (uint32_t)KP * ldg_b;

  // stage the NEXT K-step of this workgroup (K-steps are staged in order) into buffer `buf`
  auto issue_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const bool ok = s_pm[i] < kend;
      const bool okx = ok && ((unsigned)s_h[i] < (unsigned)p.hi) && ((unsigned)s_w[i] < (unsigned)p.wi);
      const uint32_t vg = ok ? s_gb[i] : kOobW;
      const uint32_t vx = okx ? s_xb[i] : kOobW;
      const uint32_t la = lds_wave + (uint32_t)(buf * (KP * RB) + i * 4096);
      const uint32_t lb = la + (uint32_t)(2 * KP * RB);
      asm volatile(
          "s_nop 4\n\t"
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %4, 0 offen lds\n\t"
          "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, 0 offen lds"
          ::"v"(vg), "v"(vx), "s"(la), "s"(lb), "s"(rs_g), "s"(rs_x)
          : "memory");
      // the slot's position KP pixels on
      s_pm[i] += KP;
      s_gb[i] += adv_g;
      int w = s_w[i] + adv_w, h = s_h[i] + adv_h;
      uint32_t xb = s_xb[i] + adv_x;
      const bool c1 = w >= lim_w;
      w -= c1 ? wrap_w : 0;
      h += c1 ? p.stride : 0;
      xb += c1 ? car_ow : 0u;
      const bool c2 = h >= lim_h;
      h -= c2 ? wrap_h : 0;
      xb += c2 ? car_oh : 0u;
      s_w[i] = w;
      s_h[i] = h;
      s_xb[i] = xb;
    }
  };

  f32x4 acc[NFR][NFR];
#pragma unroll
  for (int f = 0; f < NFR; ++f)
#pragma unroll
    for (int j = 0; j < NFR; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = WSPLIT ? 0 : (wave >> 1), wc = WSPLIT ? 0 : (wave & 1);
  const int grp = lane >> 4, i16 = lane & 15;

  auto compute = [&](int buf) {
#pragma unroll
   for (int sub0 = 0; sub0 < (WSPLIT ? 1 : KSUB); ++sub0) {
    const int sub = WSPLIT ? wave : sub0;
    const unsigned char* a = sA + buf * (KP * RB) + sub * (32 * RB);
    const unsigned char* b = sB + buf * (KP * RB) + sub * (32 * RB);
    if constexpr (sizeof(T) == 2) {
      // transposing read: lane 4q+pp of a 16-lane group addresses pixel row (4*grp + q), channels 4pp..4pp+3
      const int q = i16 >> 2, pp = i16 & 3;
      const int row = 4 * grp + q;
      const int sz = wg_swz<T, BT>(row);
      const int sub = (pp & 1) << 3;
      s16x8 af[NFR], bfr[NFR];
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        const int cb = wr * (BT / 2) + f * 16;
        const int off = row * RB + ((((cb >> 3) + (pp >> 1)) ^ sz) << 4) + sub;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + off + 16 * RB));
        af[f] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const int cb = wc * (BT / 2) + j * 16;
        const int off = row * RB + ((((cb >> 3) + (pp >> 1)) ^ sz) << 4) + sub;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + off + 16 * RB));
        bfr[j] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int f = 0; f < NFR; ++f)
#pragma unroll
        for (int j = 0; j < NFR; ++j)
          acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[f]), __builtin_bit_cast(bf16x8, bfr[j]),
                                                              acc[f][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int row = 4 * st + grp;
        const int sz = wg_swz<T, BT>(row);
        float av[NFR], bv[NFR];
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
          const int col = wr * (BT / 2) + f * 16 + i16;
          av[f] = *reinterpret_cast<const float*>(a + row * RB + (((col >> 2) ^ sz) << 4) + ((col & 3) << 2));
        }
#pragma unroll
        for (int j = 0; j < NFR; ++j) {
          const int col = wc * (BT / 2) + j * 16 + i16;
          bv[j] = *reinterpret_cast<const float*>(b + row * RB + (((col >> 2) ^ sz) << 4) + ((col & 3) << 2));
        }
#pragma unroll
        for (int f = 0; f < NFR; ++f)
#pragma unroll
          for (int j = 0; j < NFR; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f], bv[j], acc[f][j], 0, 0, 0);
      }
    }
   }
  };

  if constexpr (sizeof(T) == 2) {
    // bf16: the fragments of a whole staged K-step live in registers, double-buffered — the transposing reads of
    // tile k+1 run underneath the MFMAs of tile k and the DMA of tile k+2 (same scheme as igemm.hip)
    constexpr int NSUB = WSPLIT ? 1 : KSUB;  // 32-pixel sub-steps this wave computes per staged K-step
    typedef short s16x4v __attribute__((ext_vector_type(4)));
    s16x4v fa[2][NSUB][NFR][2], fb[2][NSUB][NFR][2];  // [set][sub-step][fragment][pixel half]
    const int q = i16 >> 2, pp = i16 & 3;
    const int row = 4 * grp + q;
    const int sz = wg_swz<T, BT>(row);
    const int sub8 = (pp & 1) << 3;
    int offa[NFR], offb[NFR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      offa[f] = row * RB + (((((wr * (BT / 2) + f * 16) >> 3) + (pp >> 1)) ^ sz) << 4) + sub8;
      offb[f] = row * RB + (((((wc * (BT / 2) + f * 16) >> 3) + (pp >> 1)) ^ sz) << 4) + sub8;
    }
#define WG_LOAD(SET, BUF)                                                                                            \
  _Pragma("unroll") for (int u_ = 0; u_ < NSUB; ++u_) {                                                              \
    const int sb_ = WSPLIT ? wave : u_;                                                                              \
    const unsigned char* a_ = sA + (BUF) * (KP * RB) + sb_ * (32 * RB);                                              \
    const unsigned char* b_ = sB + (BUF) * (KP * RB) + sb_ * (32 * RB);                                              \
    _Pragma("unroll") for (int f_ = 0; f_ < NFR; ++f_) {                                                             \
      fa[SET][u_][f_][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a_ + offa[f_]));           \
      fa[SET][u_][f_][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a_ + offa[f_] + 16 * RB)); \
      fb[SET][u_][f_][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b_ + offb[f_]));           \
      fb[SET][u_][f_][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b_ + offb[f_] + 16 * RB)); \
    }                                                                                                                \
  }
#define WG_MFMA(SET)                                                                                                 \
  _Pragma("unroll") for (int u_ = 0; u_ < NSUB; ++u_) _Pragma("unroll") for (int f_ = 0; f_ < NFR; ++f_)            \
      _Pragma("unroll") for (int j_ = 0; j_ < NFR; ++j_) {                                                           \
    const s16x8 av_ = (s16x8){fa[SET][u_][f_][0][0], fa[SET][u_][f_][0][1], fa[SET][u_][f_][0][2], fa[SET][u_][f_][0][3],   \
                              fa[SET][u_][f_][1][0], fa[SET][u_][f_][1][1], fa[SET][u_][f_][1][2], fa[SET][u_][f_][1][3]};  \
    const s16x8 bv_ = (s16x8){fb[SET][u_][j_][0][0], fb[SET][u_][j_][0][1], fb[SET][u_][j_][0][2], fb[SET][u_][j_][0][3],   \
                              fb[SET][u_][j_][1][0], fb[SET][u_][j_][1][1], fb[SET][u_][j_][1][2], fb[SET][u_][j_][1][3]};  \
    acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av_), __builtin_bit_cast(bf16x8, bv_), acc[f_][j_], 0, 0, 0); \
  }
#define WG_STEP(K, CUR, NXT)                                                                                         \
  {                                                                                                                  \
    VDQN_GST(gst_comp)                                                                                               \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                      \
    VDQN_GST(gst_wait)                                                                                               \
    _Pragma("unroll") for (int u_ = 0; u_ < NSUB; ++u_) _Pragma("unroll") for (int f_ = 0; f_ < NFR; ++f_)          \
        asm volatile("" : "+v"(fa[CUR][u_][f_][0]), "+v"(fa[CUR][u_][f_][1]), "+v"(fb[CUR][u_][f_][0]), "+v"(fb[CUR][u_][f_][1])); \
    __builtin_amdgcn_s_barrier();                                                                                    \
    VDQN_GST(gst_bar)                                                                                                \
    if ((K) + 2 < nk) issue_tile((K) & 1);                                                                           \
    VDQN_GST(gst_issue)                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    /* unconditional (the last step reads a stale buffer): one block with the MFMAs, reads issued behind them */     \
    WG_LOAD(NXT, ((K) + 1) & 1)                                                                                      \
    WG_MFMA(CUR)                                                                                                     \
    _Pragma("unroll") for (int g_ = 0; g_ < NSUB * NFR * (NFR < 4 ? NFR : 4); ++g_) {                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                                             \
      __builtin_amdgcn_sched_group_barrier(0x100, NFR < 4 ? 4 / NFR : 1, 0);                                         \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
#ifdef VDQN_STAMP
    // diagnostic build (tools/stamp_wgrad.py generic): s_memtime around the phases of every K-step, summed by thread 0
    unsigned long long gst_wait = 0, gst_bar = 0, gst_issue = 0, gst_comp = 0;
    const unsigned long long gst_begin = __builtin_amdgcn_s_memtime(), gst_rt_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long gst_t = gst_begin;
#define VDQN_GST(ACC) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ACC += n_ - gst_t; gst_t = n_; }
#else
#define VDQN_GST(ACC)
#endif
    issue_tile(0);
    if (nk > 1) {
      issue_tile(1);
      if constexpr (2 * NL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (2 * NL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    WG_LOAD(0, 0)
#ifdef VDQN_STAMP
    gst_t = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < nk; k += 2) {
      WG_STEP(k, 0, 1)
      if (k + 1 < nk) WG_STEP(k + 1, 1, 0)
    }
    VDQN_GST(gst_comp)
#ifdef VDQN_STAMP
    if (p.stamps && tid == 0) {  // (the epilogue's end is not stamped: o[2] = loop end)
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stamps) + (size_t)blockIdx.x * 16;
      o[0] = gst_begin; o[1] = gst_t; o[2] = gst_t;
      o[3] = gst_wait; o[4] = gst_bar; o[5] = gst_issue; o[6] = gst_comp; o[7] = (unsigned long long)nk;
      o[8] = gst_rt_begin; o[9] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef VDQN_GST
#undef WG_LOAD
#undef WG_MFMA
#undef WG_STEP
  } else {
    issue_tile(0);
    for (int k = 0; k < nk; ++k) {
      const int buf = k & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // tile k landed for every wave; every wave is done with tile k-1
      if (k + 1 < nk) issue_tile(buf ^ 1);
      compute(buf);
    }
  }

  // C layout: col (lane & 15) -> ci, row ((lane >> 4) * 4 + reg) -> co
  const size_t row_len = (size_t)p.taps * p.ci;
  const bool det = p.ws != nullptr;
  float* const dwp = det ? p.ws + (size_t)split * (size_t)p.ws_stride : p.dw;
  if constexpr (WSPLIT) {
    static_assert(!WSPLIT || KSUB == 4, "wave split needs 4 sub-steps");
    __syncthreads();  // everyone is done with the staging buffers: reuse them as 4 x [64][64] f32
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int f = 0; f < NFR; ++f)
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wave * 4096 + (f * 16 + grp * 4 + reg) * 64 + j * 16 + i16] = acc[f][j][reg];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int o = tid + 256 * e;
      const float v = red[o] + red[4096 + o] + red[8192 + o] + red[12288 + o];
      const int co = co0 + (o >> 6), ci = ci0 + (o & 63);
      if (co < p.co) wg_emit(dwp + (size_t)co * row_len + (size_t)tap * p.ci + ci, det, v);
    }
  } else if constexpr (sizeof(T) == 2) {
    // stage the 128x128 f32 tile through LDS so that every wave-instruction adds 256 contiguous bytes of one dw row
    // (the accumulator layout would give 4 rows x 64 B per instruction)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [BT][BT] f32 = the 64 KiB of staging buffers
#pragma unroll
    for (int f = 0; f < NFR; ++f)
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          red[(wr * (BT / 2) + f * 16 + grp * 4 + reg) * BT + wc * (BT / 2) + j * 16 + i16] = acc[f][j][reg];
    __syncthreads();
#pragma unroll 8
    for (int e = 0; e < BT * BT / 256; ++e) {
      const int o = tid + 256 * e;
      const int co = co0 + o / BT, ci = ci0 + o % BT;
      if (co < p.co) wg_emit(dwp + (size_t)co * row_len + (size_t)tap * p.ci + ci, det, red[o]);
    }
  } else {
#pragma unroll
    for (int f = 0; f < NFR; ++f)
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const int ci = ci0 + wc * (BT / 2) + j * 16 + i16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int co = co0 + wr * (BT / 2) + f * 16 + grp * 4 + reg;
          if (co < p.co) wg_emit(dwp + (size_t)co * row_len + (size_t)tap * p.ci + ci, det, acc[f][j][reg]);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Window variant for 3x3 / stride 1 / pad 1 layers (bf16): one block computes the THREE horizontal taps of a kernel row.
// The gy tile is shared by the three taps and the x operand is staged once as a window of KP + 8 consecutive pixels
// (window row j = the input pixel one row offset (kr - 1) above/below output pixel kb - 1 + j); tap ks reads it at a row
// offset of ks, and lanes whose pixel sits at the left / right image border read a zero row for ks = 0 / 2.  Per staged
// byte the block does three times the MFMAs of wgrad_kernel (which restages both operands for every tap).
// Grid: (co tile, kernel row, ci tile) x split of the pixel range.
// ---------------------------------------------------------------------------------------------------------
// 64 x 64 tiles, four waves, two workgroups per CU.  (Eight-wave 128 x 128 tiles — stride 1 and a stride-2 window form — and
// 128 x 64 tiles were built and measured slower: experiments/, DESIGN.md section 6d.  The 64 x 64 tiles of two co-resident
// workgroups ask the L2 -> LDS path for 43 B per cycle and CU at full MFMA rate, against the ~33 it delivers —
// MI355X_MICROARCH.md, gather into LDS.)
constexpr int kWgZeroBytes = 16 * 256 + 256;  // wgrad_win: zero block (covers the +16-row immediate of high-half reads for 128- and 256-byte rows)
constexpr int kWgCodeBytes = 3200;            // border-code table: images up to 56 x 56

__global__ __launch_bounds__(256, 2) void wgrad_win_kernel(const WgradParams p) {
  typedef bf16raw T;
  constexpr int BCO = 64, BCI = 64, NW = 4;
  constexpr int E16 = 8;
  constexpr int NT = 64 * NW;
  constexpr int RBG = BCO * 2, RBX = BCI * 2;      // bytes per staged gy / x row (one pixel)
  constexpr int CPRG = RBG / 16, CPRX = RBX / 16;
  // wave (wave & 1) owns ALL 64 output channels x one HALF of the input channels (three taps: 24
  // accumulator tiles = 96 VGPRs) for TWO of the four 32-pixel sub-steps of a staged K-step (waves 0, 1: sub-steps 0, 1; waves
  // 2, 3: sub-steps 2, 3); the two partial tiles of a channel half are summed through LDS before the atomics.  (Round 2 gave
  // every wave the whole 64 x 64 x 3 tile = 192 accumulator VGPRs for one sub-step: no room for a second fragment set, every
  // group of four MFMAs waited lgkmcnt(0) for its own fragment pair, and five registers spilled into the K loop.)
  constexpr int KSUB = 4;                          // 32-pixel sub-steps of a staged K-step
  constexpr int KP = 32 * KSUB;
  constexpr int NLG = (KP * CPRG) / NT;
  constexpr int WR = KP + 8;                       // window rows (KP + 2 needed)
  constexpr int HI = 16 * RBX;                     // LDS distance of a transposing read's high half (16 tile pixels further on)
  constexpr int ZBYTES = kWgZeroBytes;
  constexpr int NLX = (WR * CPRX + NT - 1) / NT;   // 16-byte slots per thread for the window (the last pass is partial)
  constexpr int NFA = 4;                           // 16-wide fragments per wave: output channels
  constexpr int NFB = 2;                           //                              input channels
  constexpr int NSUBW = 2;                         // sub-steps of a staged K-step this wave computes
  static_assert((KP * CPRG) % NT == 0 && (WR * CPRX - NT * (NLX - 1)) % 64 == 0, "staging passes are whole waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                         // [2][KP * RBG]   gy tiles
  unsigned char* sX = smem + 2 * KP * RBG;          // [2][WR * RBX]   x windows
  unsigned char* sZ = sX + 2 * WR * RBX;            // [kWgZeroBytes]  zeros at +0 and at +16 rows (every LDS bank once each; a multiple of 256 from smem)
  unsigned char* sCode = sZ + ZBYTES;               // [kWgCodeBytes]  border code of every pixel position of an image
  static_assert((2 * KP * RBG) % 256 == 0 && (WR * RBX) % 256 == 0, "window buffers and the zero block sit at 256-byte boundaries");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int n_tiles = (int)(gridDim.x / (uint32_t)p.splitk);
  const int split = (int)(lb / (uint32_t)n_tiles);
  int t = (int)(lb - (uint32_t)split * (uint32_t)n_tiles);
  const int ci_tile = t % p.ci_tiles;
  t /= p.ci_tiles;
  const int kr = t % 3;
  const int co_tile = t / 3;
  const int co0 = co_tile * BCO, ci0 = ci_tile * BCI;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.M, kbeg + p.kchunk);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + KP - 1) / KP;
  // zeros: [0, 256) for the low-half reads of border lanes and [16 RBX, 16 RBX + 256) for their high-half reads, which carry the
  // same +16-row immediate as the reads of the window (so one select serves an address pair)
  for (int i = tid; i < 32; i += NT) reinterpret_cast<uint4*>(sZ + (i >> 4) * HI)[i & 15] = make_uint4(0, 0, 0, 0);
  const int howo = (int)p.d_howo.div;
  // border code of image position rem = oh * W + ow: 1 top row, 2 bottom row, 4 left column, 8 right column.  One table per
  // workgroup instead of two divisions per pixel and K tile in every lane.
  for (int i = tid; i < howo; i += NT) {
    const uint32_t oh = fastdiv((uint32_t)i, p.d_wo), ow = (uint32_t)i - oh * p.d_wo.div;
    sCode[i] = (unsigned char)((oh == 0 ? 1u : 0u) | (oh == (uint32_t)p.ho - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == (uint32_t)p.wo - 1 ? 8u : 0u));
  }

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long g_ptr = (unsigned long long)p.gy, x_ptr = (unsigned long long)p.x;
  const i32x4 rs_g = {__builtin_amdgcn_readfirstlane((int)(unsigned)g_ptr), __builtin_amdgcn_readfirstlane((int)((g_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.gy_bytes), 0x00020000};
  const i32x4 rs_x = {__builtin_amdgcn_readfirstlane((int)(unsigned)x_ptr), __builtin_amdgcn_readfirstlane((int)((x_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.x_bytes), 0x00020000};
  // 16-byte slot q = tid + 256 i of a staged tile: row q / CPR, source chunk (q % CPR) ^ swizzle(row) — recomputed per
  // issue (shifts and xors) instead of held in registers: the three accumulator tiles need most of the VGPRs
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * 1024u;
  const uint32_t ldg_b = (uint32_t)p.ldg * 2u;
  const uint32_t pix_b = (uint32_t)p.pix_stride * 2u;
  constexpr int X_TAIL_WAVES = (WR * CPRX - NT * (NLX - 1)) / 64;  // waves that take part in the last window pass

  // Staging addresses are LINEAR in the pixel index: gy row pm, and — stride 1, pad 1, same-size images laid end to end — the
  // x pixel one kernel-row offset away is pixel pm + (kr - 1) W of the flat tensor.  A pixel index below 0 or past the tensor
  // wraps / runs out of the buffer descriptor's range and is zero-filled by the hardware; pixels that are in range but belong to
  // another image row (top / bottom / left / right border of the OUTPUT pixel) are handled where the fragments are read (zero
  // row).  So a K-step's address arithmetic is one add per DMA piece (it was ~25 VALU per piece with two divisions).
  // slot q = tid + 256 i of a staged tile is row q / CPR + (256 / CPR) i with the SAME source chunk for every i (the swizzle keys
  // repeat every 8 rows), so one per-lane offset per operand serves all pieces and the piece's row block is a scalar
  const uint32_t g_lane = (uint32_t)(tid / CPRG) * ldg_b + (uint32_t)((co0 + ((tid % CPRG) ^ wg_swz<T, BCO>(tid / CPRG)) * E16) * 2);
  const int x_key = wg_swz<T, BCI>(tid / CPRX);
  const uint32_t x_chunk = (uint32_t)((ci0 + ((tid % CPRX) ^ x_key) * E16) * 2);
  const uint32_t x_lane = (uint32_t)(tid / CPRX) * pix_b + x_chunk;
  static_assert((NT / CPRG) % 8 == 0 && (NT / CPRX) % 8 == 0, "swizzle keys repeat over the pieces of one thread");
  const int x_shift = (kr - 1) * p.wo - 1;  // window row j = pixel kb + j + x_shift

  // one LDS-DMA piece of the tile at pixel kb: pieces 0 .. NLG - 1 = gy rows, NLG .. NLG + NLX - 1 = window rows
  auto issue_piece = [&](int kb, int buf, int i) {
    if (i < NLG) {
      const uint32_t vg = g_lane + (uint32_t)(kb + (NT / CPRG) * i) * ldg_b;
      const uint32_t la = lds_wave + (uint32_t)(buf * (KP * RBG) + i * (NT * 16));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(vg), "s"(la), "s"(rs_g) : "memory");
    } else {
      const int ix = i - NLG;
      if (ix == NLX - 1 && wave_u >= X_TAIL_WAVES) return;
      const uint32_t vx = x_lane + (uint32_t)(kb + x_shift + (NT / CPRX) * ix) * pix_b;  // may wrap below zero: out of range, zero-filled
      const uint32_t lx = lds_wave + (uint32_t)(2 * KP * RBG + buf * (WR * RBX) + ix * (NT * 16));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(vx), "s"(lx), "s"(rs_x) : "memory");
    }
  };
  auto issue_tile = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < NLG + NLX; ++i) issue_piece(kb, buf, i);
  };

  f32x4 acc[3][NFA][NFB];
#pragma unroll
  for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
    for (int f = 0; f < NFA; ++f)
#pragma unroll
      for (int j = 0; j < NFB; ++j) acc[k3][f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wc = wave & 1;
  const int grp = lane >> 4, i16 = lane & 15;
  const int q4 = i16 >> 2, pp = i16 & 3;
  const int row = 4 * grp + q4;            // pixel row (of 32) this lane addresses in a transposing read; + 16 for the high half
  const int sub8 = (pp & 1) << 3;
  const int szA = wg_swz<T, BCO>(row);

  // ---- everything a fragment read needs is a per-lane constant of the workgroup, computed here once ----
  // The K loop of round 2 issued ~4.8 vector instructions per MFMA (SQ_INSTS_VALU / SQ_INSTS_MFMA = 6 over the whole kernel): LDS
  // addresses rebuilt for every read, two divisions per pixel for the border test, selects — at 4 issue cycles each beside 8 per
  // MFMA that is 1300 issue cycles per wave and K tile for 768 cycles of matrix work, with two waves per SIMD: the loop was bound
  // by VALU issue, not by LDS, the L2 -> LDS fill or the atomics (profiles/r03d_pmc_mfma.json; the experiments that moved those
  // three are in experiments/README.md).  Now: xo[i][ks][j] = LDS offset (inside a window buffer) of the x fragment of the wave's
  // sub-step i, tap ks, channel fragment j; go[f] = offset of gy fragment f inside a sub-step's 32 gy rows; the buffer index is a
  // compile-time constant (the K loop is unrolled by two), so buffer / sub-step / high-half offsets are ds_read immediates; a
  // border lane's read pair is redirected by ONE select pair to the zero block at its own bank position (off & 255).
  const int sub0 = (wave >> 1) * 2;  // first sub-step of this wave
  uint32_t xo[NSUBW][3][NFB], go[NFA];
  uint32_t rem[NSUBW][2];  // position inside its image (oh * W + ow) of the wave's pixels (sub-step i, half hh) of the CURRENT K tile
  bool zb[NSUBW][3][2];    // zb[i][ks][hh]: tap ks of kernel row kr leaves the image for that pixel (lane masks in scalar registers)
  const uint32_t lds_smem = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t z_base = lds_smem + (uint32_t)(sZ - smem);
  // border code bits that put the whole kernel row outside
  const uint32_t vmask = kr == 0 ? 1u : (kr == 2 ? 2u : 0u);
  const uint32_t kp_mod = (uint32_t)(KP % howo);
  auto masks_from_rem = [&](int i) {
    uint32_t c[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) c[hh] = sCode[rem[i][hh]];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      zb[i][0][hh] = (c[hh] & (vmask | 4u)) != 0u;  // kernel row outside, or left column with tap 0
      zb[i][1][hh] = (c[hh] & vmask) != 0u;
      zb[i][2][hh] = (c[hh] & (vmask | 8u)) != 0u;  // ... or right column with tap 2
    }
  };
  auto advance_rem = [&](int i) {  // the same pixels of the next K tile: KP positions further on, modulo the image
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      uint32_t r = rem[i][hh] + kp_mod;
      r -= r >= (uint32_t)howo ? (uint32_t)howo : 0u;
      rem[i][hh] = r;
    }
  };
#pragma unroll
  for (int f = 0; f < NFA; ++f) {
    const int cb = f * 16;
    go[f] = lds_smem + (uint32_t)((sub0 * 32 + row) * RBG + ((((cb >> 3) + (pp >> 1)) ^ szA) << 4) + sub8);
  }
#pragma unroll
  for (int i = 0; i < NSUBW; ++i) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int jr = (sub0 + i) * 32 + row + ks;
      const int szX = wg_swz<T, BCI>(jr);
#pragma unroll
      for (int j = 0; j < NFB; ++j) {
        const int cb = wc * (16 * NFB) + j * 16;
        xo[i][ks][j] = lds_smem + (uint32_t)(2 * KP * RBG + jr * RBX + ((((cb >> 3) + (pp >> 1)) ^ szX) << 4) + sub8);
      }
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const uint32_t pm = (uint32_t)(kbeg + (sub0 + i) * 32 + row + 16 * hh);
      rem[i][hh] = pm - fastdiv(pm, p.d_howo) * p.d_howo.div;
    }
  }
  __syncthreads();  // the code table is complete
#pragma unroll
  for (int i = 0; i < NSUBW; ++i) masks_from_rem(i);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto lds_tr = [](uint32_t addr, int imm) -> s16x4 {  // ds_read_b64_tr_b16 addr offset:imm
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr) + imm / 8);
  };

  // kb_next >= 0 (-DVDQN_WGRAD_TRICKLE builds only): the LDS-DMA pieces of the NEXT tile (into the other buffer pair, free since
  // this tile's barrier) are issued one per stage BEHIND the stage's MFMAs instead of all in front of the tile's first fragment
  // read: a piece holds the wave at issue for ~65 cycles (tools/stamp_wgrad.py: 560-675 cycles per tile for nine pieces).
  // Measured: 1.00 against 0.93 ms per update for the window weight gradients — the pieces stall the MFMA stream more than
  // they stall in front of it.  Not the default.
  auto compute = [&](int buf, int kb_next) {
    {
      // Software pipeline over the wave's NSUBW sub-steps: NSUBW x (3 taps x NFB channel fragments) stages of NFA MFMAs (one x
      // fragment against the NFA gy fragments).  The x fragment of stage s + 2 is read while stage s computes (three register
      // sets), the gy fragments of the next sub-step (second register set) while the current one's stages 2.. compute: no MFMA
      // waits for an LDS read it has just issued.
      constexpr int SPS = 3 * NFB;           // stages per sub-step
      constexpr int NST = NSUBW * SPS;
      static_assert(NFB == 2 && NFA == 4 && SPS >= 2 + NFA, "stage layout: the next sub-step's four gy fragments load in stages 2..5");
      // (wave-uniform buffer offsets: one v_add with a scalar operand per address; sub-step and high-half offsets are immediates)
      const uint32_t xbuf = (uint32_t)__builtin_amdgcn_readfirstlane(buf * (WR * RBX)), gbuf = (uint32_t)__builtin_amdgcn_readfirstlane(buf * (KP * RBG));
      s16x8 af[2][NFA], bj[3];
      auto read_g = [&](int sbi, int f) -> s16x8 {
        const uint32_t a = go[f] + gbuf;
        const s16x4 lo = lds_tr(a, sbi * (32 * RBG)), hi = lds_tr(a, sbi * (32 * RBG) + 16 * RBG);
        return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      };
      auto stage_x = [&](int st) -> s16x8 {  // st = sub-step * SPS + ks * NFB + j
        const int sb = st / SPS, ks = (st % SPS) / NFB, j = st % NFB;
        const uint32_t off = xo[sb][ks][j];
        const uint32_t xaddr = off + xbuf;
        const uint32_t zaddr = z_base | ((off - lds_smem) & 255u);  // the lane's own bank position inside the zero block (z_base is a multiple of 256)
        const uint32_t a0 = zb[sb][ks][0] ? zaddr : xaddr, a1 = zb[sb][ks][1] ? zaddr : xaddr;
        const s16x4 lo = lds_tr(a0, 0), hi = lds_tr(a1, HI);
        return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      };
#pragma unroll
      for (int f = 0; f < NFA; ++f) af[0][f] = read_g(0, f);
      bj[0] = stage_x(0);
      bj[1] = stage_x(1);
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int sb = st / SPS, ss = st % SPS, ks = ss / NFB, j = ss % NFB;
        if (st + 2 < NST) bj[(st + 2) % 3] = stage_x(st + 2);
        // the next sub-step's gy fragments, one per stage (its register set was last read by the previous sub-step's final stage)
        if (sb + 1 < NSUBW && ss >= 2 && ss < 2 + NFA) af[(sb + 1) & 1][ss - 2] = read_g(sb + 1, ss - 2);
        // the NEXT tile's border masks of sub-step sb - 1: its current masks were last used by a read issued three stages ago
        if (ss == 1 && sb > 0) advance_rem(sb - 1);
        if (ss == 3 && sb > 0) masks_from_rem(sb - 1);
#pragma unroll
        for (int f = 0; f < NFA; ++f)
          acc[ks][f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[sb & 1][f]), __builtin_bit_cast(bf16x8, bj[st % 3]), acc[ks][f][j], 0, 0, 0);
        // pin the stage order: left to itself the scheduler sinks every read to just in front of its first use and waits
        // lgkmcnt(0) there.  With a scheduling fence per stage the reads stay two stages ahead and the compiler's own counted
        // lgkmcnt waits (it tracks the builtin reads) let each stage start as soon as ITS fragment has landed.
        __builtin_amdgcn_sched_barrier(0);
#ifdef VDQN_WGRAD_TRICKLE
        if (st < NLG + NLX && kb_next >= 0) {
          issue_piece(kb_next, buf ^ 1, st);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      }
      static_assert(NLG + NLX <= NST, "one LDS-DMA piece per stage");
      advance_rem(NSUBW - 1);  // (after the last stage: the final sub-step's masks were in use until stage NST - 3)
      masks_from_rem(NSUBW - 1);
    }
  };

#ifdef VDQN_STAMP
  unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_comp = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime(), st_rt_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_t = st_begin;
#define VDQN_WST(ACC) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ACC += n_ - st_t; st_t = n_; }
#else
#define VDQN_WST(ACC)
#endif
  issue_tile(kbeg, 0);
  VDQN_WST(st_issue)
  for (int k = 0; k < nk; ++k) {
    const int buf = k & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VDQN_WST(st_wait)
    __builtin_amdgcn_s_barrier();  // tile k landed for every wave; every wave is done with tile k-1
    VDQN_WST(st_bar)
#ifdef VDQN_WGRAD_TRICKLE  // (build flag: measured 8 % slower per kernel, profiles/r03h_ab_wgrad_trickle.txt — off)
    constexpr bool kTrickle = true;
#else
    constexpr bool kTrickle = false;
#endif
    if (!kTrickle && k + 1 < nk) issue_tile(kbeg + (k + 1) * KP, buf ^ 1);
    VDQN_WST(st_issue)
    compute(buf, (kTrickle && k + 1 < nk) ? kbeg + (k + 1) * KP : -1);
    VDQN_WST(st_comp)
  }
#ifdef VDQN_STAMP
  const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif

  // ---- three tap tiles -> dw (f32 atomics, or plain stores into this split's copy; contiguous runs through LDS) ----
  const size_t row_len = (size_t)p.taps * p.ci;
  const bool det = p.ws != nullptr;
  float* const dwp = det ? p.ws + (size_t)split * (size_t)p.ws_stride : p.dw;
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int tap = kr * 3 + ks;
    __syncthreads();  // staging buffers (or the previous tap's tile) are free
#pragma unroll
    for (int f = 0; f < NFA; ++f)
#pragma unroll
      for (int j = 0; j < NFB; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[(wave >> 1) * 4096 + (f * 16 + grp * 4 + reg) * 64 + wc * 32 + j * 16 + i16] = acc[ks][f][j][reg];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int o = tid + 256 * e;
      const float v = red[o] + red[4096 + o];
      const int co = co0 + (o >> 6), ci = ci0 + (o & 63);
      if (co < p.co) wg_emit(dwp + (size_t)co * row_len + (size_t)tap * p.ci + ci, det, v);
    }
  }
#ifdef VDQN_STAMP
  if (p.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stamps) + (size_t)blockIdx.x * 16;
    o[0] = st_begin; o[1] = st_loop_end; o[2] = __builtin_amdgcn_s_memtime();
    o[3] = st_wait; o[4] = st_bar; o[5] = st_issue; o[6] = st_comp; o[7] = (unsigned long long)nk;
    o[8] = st_rt_begin; o[9] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#undef VDQN_WST
}

// dbias[c] += sum_m gy[m][c]: 16-byte column groups x row stripes per block, stripes reduced through LDS,
// one atomic per column per block (<= 256 blocks).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ gy, float* __restrict__ dbias, int M, int C, int ldg,
                                                     int rows_per_block) {
  constexpr int E16 = 16 / (int)sizeof(T);
  __shared__ float red[256 * E16];
  const int cg_n = C / E16;            // 16-byte column groups (<= 256, power-of-two multiples of 8)
  const int nstripe = 256 / cg_n;      // row stripes per block
  const int cg = threadIdx.x % cg_n, stripe = threadIdx.x / cg_n;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s[E16];
#pragma unroll
  for (int e = 0; e < E16; ++e) s[e] = 0.f;
  if (stripe < nstripe) {
    for (int r = r0 + stripe; r < r1; r += nstripe) {
      const uint4 v = *reinterpret_cast<const uint4*>(gy + (size_t)r * ldg + cg * E16);
      const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int e = 0; e < E16; ++e) s[e] += to_f32<T>(pv[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < E16; ++e) red[threadIdx.x * E16 + e] = s[e];
  __syncthreads();
  // thread t < C sums column t over the stripes
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / E16, e = c % E16;
    float t = 0.f;
    for (int st = 0; st < nstripe; ++st) t += red[(st * cg_n + g) * E16 + e];
    atomicAdd(dbias + c, t);
  }
}

template <typename T, int BT>
int launch_wgrad(const WgradParams& p, int tiles, int splitk, hipStream_t stream) {
  const size_t smem = 4 * 32 * wg_ksub<T, BT>() * BT * sizeof(T);
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&wgrad_kernel<T, BT>), (size_t)smem);
  const double esz = sizeof(T);
  vdqn_prof_begin(sizeof(T) == 2 ? (BT == 128 ? "wgrad<bf16,128>" : "wgrad<bf16,64>") : (BT == 128 ? "wgrad<f32,128>" : "wgrad<f32,64>"),
                  2.0 * p.M * p.co * p.taps * p.ci,
                  esz * ((double)p.M * p.ldg + (double)p.n_img * p.hi * p.wi * p.ci) + 4.0 * p.co * p.taps * p.ci, stream);
  hipLaunchKernelGGL((wgrad_kernel<T, BT>), dim3(tiles * splitk), dim3(256), smem, stream, p);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the stem (conv1 as the 4x4/1 convolution over the packed space-to-depth frame, bf16):
//   dw[co][ky][kx*16 + c] += sum over conv pixels (y, x) of g_c1[y][x][co] * packed[y + ky][x + kx][c]
// The generic kernel treats every ky as a tap with a 64-wide "channel" dimension (4 packed pixels x 16) and stages, per 64
// pixels and tap, an 8 KiB gradient tile and an 8 KiB input tile whose rows overlap by 96 of 128 bytes: 32 FLOP per staged
// byte, i.e. bound by the L2 -> LDS fill rate (~420 TFLOP/s on the padded K).  Here a workgroup walks whole conv rows: per
// row it stages the 112 x 64 gradient tile ONCE for all four ky and the four packed input rows y .. y+3 as they lie in memory
// (115 x 32 bytes each, no duplication); wave ky reads its B fragments at a shift of kx packed pixels.  Rows are padded to
// 128 pixels with gradient rows that stay zero (4 x 32-pixel MFMA sub-steps, 12.5 % padding).  Partial 64 x 256 results are
// added to dw with f32 atomics like the generic kernel's.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSwGyBytes = 128 * 128;             // [128 pixel rows][64 co] bf16; rows 112.. are never staged and stay zero
constexpr int kSwXRow = 4096 + 128;               // one packed input row: 128 px x 32 B (115 real) + a zero tail for the kx shift
constexpr int kSwBuf = kSwGyBytes + 4 * kSwXRow;  // 33280 bytes
constexpr int kSwSmem = 2 * kSwBuf;

__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const bf16raw* __restrict__ gy, const bf16raw* __restrict__ x, float* __restrict__ dw,
                                                            int total_rows, int rows_per_block, int gy_bytes, int x_bytes, float* __restrict__ ws) {
  using T = bf16raw;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ky = __builtin_amdgcn_readfirstlane(wave);  // wave = kernel row
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(total_rows, r0 + rows_per_block);
  if (r0 >= r1) return;
  // the parts of both buffers the DMA never writes: gradient rows 112..127 and the tail of every packed row
  for (int i = tid; i < 2 * (128 + 32); i += 256) {
    const int b = i / 160, k = i - b * 160;
    unsigned char* d = k < 128 ? smem + b * kSwBuf + 112 * 128 + k * 16 : smem + b * kSwBuf + kSwGyBytes + ((k - 128) >> 3) * kSwXRow + 4096 + ((k - 128) & 7) * 16;
    *reinterpret_cast<uint4*>(d) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long g_ptr = (unsigned long long)gy, x_ptr = (unsigned long long)x;
  const i32x4 rs_g = {__builtin_amdgcn_readfirstlane((int)(unsigned)g_ptr), __builtin_amdgcn_readfirstlane((int)((g_ptr >> 32) & 0xffff)), gy_bytes, 0x00020000};
  const i32x4 rs_x = {__builtin_amdgcn_readfirstlane((int)(unsigned)x_ptr), __builtin_amdgcn_readfirstlane((int)((x_ptr >> 32) & 0xffff)), x_bytes, 0x00020000};
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // per-lane parts of the source offsets: gradient piece = 8 pixel rows x 128 B (source chunk XOR-swizzled like wgrad_kernel's
  // 64-wide tiles), input piece = 32 packed pixels x 32 B
  const int g_jj = lane >> 3;
  const uint32_t x_lane = (uint32_t)((lane >> 1) * 32 + (lane & 1) * 16);

  auto issue_row = [&](int R, int buf) {
    const int img = R / 112, y = R - img * 112;
    const uint32_t g_row0 = (uint32_t)(img * 12544 + y * 112);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int piece = ky + 4 * k;  // wave-uniform
      if (piece < 14) {
        const int row = piece * 8 + g_jj;
        const uint32_t vg = (g_row0 + (uint32_t)row) * 128u + (uint32_t)((((lane & 7) ^ wg_swz<T, 64>(row)) << 4));
        const uint32_t l_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + (uint32_t)(buf * kSwBuf + piece * 1024)));
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(vg), "s"(l_), "s"(rs_g) : "memory");
      }
    }
    const uint32_t x_row0 = (uint32_t)((img * 115 + y + ky) * 115) * 32u + x_lane;
    const uint32_t lx = lds_base + (uint32_t)(buf * kSwBuf + kSwGyBytes + ky * kSwXRow);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t vx = x_row0 + (uint32_t)(j * 1024);
      const uint32_t l_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lx + (uint32_t)(j * 1024)));
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(vx), "s"(l_), "s"(rs_x) : "memory");
    }
  };

  f32x4 acc[4][4];  // [co fragment][kx]
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int grp = lane >> 4, i16 = lane & 15;
  const int q = i16 >> 2, pp = i16 & 3;
  const int row = 4 * grp + q;  // pixel row (of 32) this lane addresses in a transposing read; the high half is 16 rows on
  const int sz = wg_swz<T, 64>(row);
  int offa[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) offa[f] = row * 128 + (((((f * 16) >> 3) + (pp >> 1)) ^ sz) << 4) + ((pp & 1) << 3);
  const int offb = row * 32 + pp * 8;

  issue_row(r0, 0);
  int buf = 0;
  for (int R = r0; R < r1; ++R, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // row R landed for every wave; every wave is done with the other buffer
    if (R + 1 < r1) issue_row(R + 1, buf ^ 1);
    const unsigned char* a = smem + buf * kSwBuf;
    const unsigned char* b = smem + buf * kSwBuf + kSwGyBytes + ky * kSwXRow;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      s16x8 af[4], bfr[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const unsigned char* pa = a + sub * (32 * 128) + offa[f];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 16 * 128));
        af[f] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const unsigned char* pb = b + (sub * 32 + kx) * 32 + offb;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pb);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + 16 * 32));
        bfr[kx] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
          acc[f][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[f]), __builtin_bit_cast(bf16x8, bfr[kx]), acc[f][kx], 0, 0, 0);
    }
  }
  // C layout: col (lane & 15) -> kx*16 + c, row ((lane >> 4) * 4 + reg) -> co
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        wg_emit((ws ? ws + (size_t)blockIdx.x * (64 * 256) : dw) + (size_t)(f * 16 + grp * 4 + reg) * 256 + ky * 64 + kx * 16 + i16, ws != nullptr, acc[f][kx][reg]);
}

// ---------------------------------------------------------------------------------------------------------
// The stem's weight gradient straight from the POOLED gradient (bf16): max-pool backward fused into the staging of
// stem_wgrad_kernel's gradient operand.  Unfused, the 112 x 112 x 64 gradient of conv1's output — three quarters of it zeros —
// is written by vdqn_maxpool_bwd (411 MB at batch 256) and read back by the weight-gradient kernel: the two HBM-bound launches
// at the very end of an update's backward chain.  Here a workgroup walks pooled rows k of one image: conv rows 2k and 2k + 1
// receive gradient only from pooled rows k and k + 1 (3 x 3 / stride 2 / pad 1 windows), which sit in LDS (gradient + arg-max
// codes, one new row per step by LDS-DMA); the two 112 x 64 gradient tiles are built from them with the arithmetic of
// maxpool_bwd_rows_kernel (f32 sum of the <= 4 windows whose arg-max is the pixel, one rounding: the tiles are bit-identical to
// the unfused tensor) and consumed by the MFMA loop of stem_wgrad_kernel (wave ky, packed input rows 2k + hr + ky, kx shifts).
// Per step and workgroup: [A] barrier | packed-row DMA issued, tiles built (VALU) | [B] barrier | next pooled row's DMA issued,
// 128 MFMAs per wave.  Two workgroups per CU: one's VALU phase runs beside the other's MFMA phase.
// HBM per update: pooled gradient + codes + packed frames = 231 MB instead of 1.27 GB for the two launches.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSpPoolG = 56 * 128;                 // a pooled gradient row: 56 pixels x 64 channels bf16 (7 DMA pieces)
constexpr int kSpPoolI = 4096;                     // its arg-max codes, 56 x 64 bytes, staged as 4 pieces (the last one half used)
constexpr int kSpPoolRow = kSpPoolG + kSpPoolI;
constexpr int kSpGy = 2 * kSwGyBytes;              // gradient tiles of conv rows 2k, 2k + 1
constexpr int kSpX = 5 * kSwXRow;                  // packed input rows 2k .. 2k + 4
constexpr int kSpSmem = 2 * kSpPoolRow + kSpGy + kSpX;  // 76416 bytes: two workgroups per CU

__global__ __launch_bounds__(256, 2) void stem_wgrad_pool_kernel(const bf16raw* __restrict__ g_pool, const uint8_t* __restrict__ idx,
                                                                 const bf16raw* __restrict__ x, float* __restrict__ dw, int pairs_per_block,
                                                                 int blocks_per_img, int gp_bytes, int idx_bytes, int x_bytes, float* __restrict__ ws,
                                                                 void* stamps) {
  using T = bf16raw;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sPool = smem;                       // [2][kSpPoolRow]: pooled row r lives in slot r & 1
  unsigned char* sGy = smem + 2 * kSpPoolRow;        // [2][kSwGyBytes]
  unsigned char* sX = sGy + kSpGy;                   // [5][kSwXRow]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ky = __builtin_amdgcn_readfirstlane(wave);  // wave = kernel row in the MFMA phase
  const int img = blockIdx.x / blocks_per_img;
  const int k0 = (blockIdx.x - img * blocks_per_img) * pairs_per_block;
  const int k1 = min(56, k0 + pairs_per_block);
  if (k0 >= k1) return;
  // what the DMA / the builder never write: gradient rows 112..127 of both tiles, the tail of every packed row; and the pooled-row
  // slots, so that a slot no row was ever staged into (pooled row 56) holds arg-max codes of 0, not whatever LDS held
  for (int i = tid; i < 2 * 128 + 5 * 8; i += 256) {
    unsigned char* d = i < 256 ? sGy + (i >> 7) * kSwGyBytes + 112 * 128 + (i & 127) * 16 : sX + ((i - 256) >> 3) * kSwXRow + 4096 + ((i - 256) & 7) * 16;
    *reinterpret_cast<uint4*>(d) = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < 2 * kSpPoolRow / 16; i += 256) reinterpret_cast<uint4*>(sPool)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();  // before the first LDS-DMA lands in a slot

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long g_ptr = (unsigned long long)g_pool, i_ptr = (unsigned long long)idx, x_ptr = (unsigned long long)x;
  const i32x4 rs_g = {__builtin_amdgcn_readfirstlane((int)(unsigned)g_ptr), __builtin_amdgcn_readfirstlane((int)((g_ptr >> 32) & 0xffff)), gp_bytes, 0x00020000};
  const i32x4 rs_i = {__builtin_amdgcn_readfirstlane((int)(unsigned)i_ptr), __builtin_amdgcn_readfirstlane((int)((i_ptr >> 32) & 0xffff)), idx_bytes, 0x00020000};
  const i32x4 rs_x = {__builtin_amdgcn_readfirstlane((int)(unsigned)x_ptr), __builtin_amdgcn_readfirstlane((int)((x_ptr >> 32) & 0xffff)), x_bytes, 0x00020000};
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t lane16 = (uint32_t)lane * 16u;
#define VDQN_SP_DMA(VOFF, LDS, RSRC)                                                                                   \
  {                                                                                                                    \
    const uint32_t l_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(LDS));                                           \
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(VOFF), "s"(l_), "s"(RSRC) : "memory"); \
  }
  // pooled row r of this image -> slot r & 1: 7 gradient pieces + 4 code pieces of 1 KiB, piece p by wave p & 3
  auto issue_pool = [&](int r) {
    if (r >= 56) return;
    const uint32_t px0 = (uint32_t)((img * 56 + r) * 56);
    const uint32_t slot = lds_base + (uint32_t)((r & 1) * kSpPoolRow);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int pc = ky + 4 * i;  // wave-uniform
      if (pc < 7) {
        VDQN_SP_DMA(px0 * 128u + (uint32_t)(pc * 1024) + lane16, slot + (uint32_t)(pc * 1024), rs_g)
      } else if (pc < 11) {
        VDQN_SP_DMA(px0 * 64u + (uint32_t)((pc - 7) * 1024) + lane16, slot + (uint32_t)(kSpPoolG + (pc - 7) * 1024), rs_i)
      }
    }
  };
  // packed input rows 2k .. 2k + 4 (each 128 packed pixels x 32 B; 115 are real, the rest runs into the next row and meets
  // gradient rows that are zero): 20 pieces, piece p by wave p & 3
  auto issue_x = [&](int k) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int pc = ky + 4 * i;
      const int row = pc >> 2, j = pc & 3;
      VDQN_SP_DMA((uint32_t)((img * 115 + 2 * k + row) * 115) * 32u + (uint32_t)(j * 1024) + lane16, lds_base + (uint32_t)(2 * kSpPoolRow + kSpGy + row * kSwXRow + j * 1024), rs_x)
    }
  };
#undef VDQN_SP_DMA
  // gradient tiles of conv rows 2k (hr = 0) and 2k + 1 (hr = 1).  An item is (conv row, pixel w, 8-channel group); row 2k only
  // receives from pooled row k (window row kh = 1), row 2k + 1 from pooled rows k (kh = 2) and k + 1 (kh = 0); an even pixel
  // w = 2 j lies in window column j only (dx = 1), an odd one in columns j (dx = 2) and j + 1 (dx = 0).  The 4 x 448 items are
  // dealt to the waves in runs of 64 of ONE (row, pixel parity) class, so the candidate windows of a run — 1, 2, 2 or 4, each with
  // a constant arg-max code to match — are the same for every lane: 7 runs per wave and step.  Sums in f32 in the order of
  // maxpool_bwd_rows_kernel (pooled row, then window column), one rounding.
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  auto add8 = [&](float (&sacc)[8], const unsigned char* slot, int ow, int cg, uint32_t tap) {
    const uint4 gv = *reinterpret_cast<const uint4*>(slot + ow * 128 + cg * 16);
    const uint2 iv = *reinterpret_cast<const uint2*>(slot + kSpPoolG + ow * 64 + cg * 8);
    const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
    const uint32_t xw[2] = {iv.x ^ (tap * 0x01010101u), iv.y ^ (tap * 0x01010101u)};  // a zero byte = a channel whose arg-max is this tap
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
      // the two code bytes of channels 2 pq, 2 pq + 1 spread to 16-bit halves; min(., 1) - 1 = 0xffff where the byte is zero
      const uint32_t t = __builtin_amdgcn_perm(0u, xw[pq >> 1], (pq & 1) ? 0x0c030c02u : 0x0c010c00u);
      const u16x2 m = __builtin_elementwise_min(__builtin_bit_cast(u16x2, t), (u16x2){1, 1}) - (u16x2){1, 1};
      const uint32_t mg = gw[pq] & __builtin_bit_cast(uint32_t, m);
      sacc[2 * pq] += __uint_as_float(mg << 16);
      sacc[2 * pq + 1] += __uint_as_float(mg & 0xffff0000u);
    }
  };
  auto build = [&](int k) {
    const bool two = k + 1 < 56;
    const unsigned char* s0 = sPool + (k & 1) * kSpPoolRow;        // pooled row k
    const unsigned char* s1 = sPool + ((k + 1) & 1) * kSpPoolRow;  // pooled row k + 1
#pragma unroll 1
    for (int i = 0; i < 7; ++i) {
      const int run = ky + 4 * i;   // wave-uniform, 0..27
      const int cls = run / 7;      // 0: row 2k, even w | 1: row 2k, odd w | 2: row 2k + 1, even w | 3: row 2k + 1, odd w
      const int j = (run - cls * 7) * 64 + lane;
      const int cg = j & 7, jp = j >> 3;
      // a window that does not exist (column 56, pooled row 56) is read at a clamped address against a code no arg-max has:
      // no branches inside a run, so its LDS reads are all in flight before the first is used
      const int jr = min(jp + 1, 55);
      const uint32_t no = 0xffu;
      const bool right = jp + 1 < 56;
      float sacc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sacc[e] = 0.f;
      if (cls == 0) {
        add8(sacc, s0, jp, cg, 4);
      } else if (cls == 1) {
        add8(sacc, s0, jp, cg, 5);
        add8(sacc, s0, jr, cg, right ? 3u : no);
      } else if (cls == 2) {
        add8(sacc, s0, jp, cg, 7);
        add8(sacc, s1, jp, cg, two ? 1u : no);
      } else {
        add8(sacc, s0, jp, cg, 8);
        add8(sacc, s0, jr, cg, right ? 6u : no);
        add8(sacc, s1, jp, cg, two ? 2u : no);
        add8(sacc, s1, jr, cg, (two && right) ? 0u : no);
      }
      const int w = 2 * jp + (cls & 1);
      T ov[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = from_f32<T>(sacc[e]);
      *reinterpret_cast<uint4*>(sGy + (cls >> 1) * kSwGyBytes + w * 128 + ((cg ^ wg_swz<T, 64>(w)) << 4)) = *reinterpret_cast<const uint4*>(ov);
    }
  };

  f32x4 acc[4][4];  // [co fragment][kx]
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int grp = lane >> 4, i16 = lane & 15;
  const int q = i16 >> 2, pp = i16 & 3;
  const int row = 4 * grp + q;  // pixel row (of 32) this lane addresses in a transposing read; the high half is 16 rows on
  const int sz = wg_swz<T, 64>(row);
  int offa[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) offa[f] = row * 128 + (((((f * 16) >> 3) + (pp >> 1)) ^ sz) << 4) + ((pp & 1) << 3);
  const int offb = row * 32 + pp * 8;

#ifdef VDQN_STAMP
  // diagnostic build (tools/stamp_wgrad.py stem): per step — wait + barrier [A] | packed-row DMA issue + tile build + its waits |
  // barrier [B] | pooled-row DMA issue + 128 MFMAs per wave (issue side)
  unsigned long long pst_a = 0, pst_build = 0, pst_b = 0, pst_mfma = 0;
  const unsigned long long pst_begin = __builtin_amdgcn_s_memtime(), pst_rt_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long pst_t = pst_begin;
#define VDQN_PST(ACC) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ACC += n_ - pst_t; pst_t = n_; }
#else
#define VDQN_PST(ACC)
#endif
  issue_pool(k0);
  issue_pool(k0 + 1);
  for (int k = k0; k < k1; ++k) {
    VDQN_PST(pst_mfma)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // [A] pooled rows k, k + 1 are in LDS; every wave is done with the tiles and the packed rows of step k - 1
    VDQN_PST(pst_a)
    issue_x(k);
    build(k);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    VDQN_PST(pst_build)
    __builtin_amdgcn_s_barrier();  // [B] tiles built, packed rows landed; pooled row k is no longer read
    VDQN_PST(pst_b)
    if (k + 1 < k1) issue_pool(k + 2);
#pragma unroll 1
    for (int hr = 0; hr < 2; ++hr) {
      const unsigned char* a = sGy + hr * kSwGyBytes;
      const unsigned char* b = sX + (hr + ky) * kSwXRow;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        s16x8 af[4], bfr[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const unsigned char* pa = a + sub * (32 * 128) + offa[f];
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 16 * 128));
          af[f] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const unsigned char* pb = b + (sub * 32 + kx) * 32 + offb;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pb);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + 16 * 32));
          bfr[kx] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int kx = 0; kx < 4; ++kx)
            acc[f][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[f]), __builtin_bit_cast(bf16x8, bfr[kx]), acc[f][kx], 0, 0, 0);
      }
    }
  }
  VDQN_PST(pst_mfma)
#ifdef VDQN_STAMP
  if (stamps && tid == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(stamps) + (size_t)blockIdx.x * 16;
    o[0] = pst_begin; o[1] = pst_t; o[2] = pst_t;
    o[3] = pst_a; o[4] = pst_b; o[5] = pst_build; o[6] = pst_mfma; o[7] = (unsigned long long)(k1 - k0);
    o[8] = pst_rt_begin; o[9] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#undef VDQN_PST
  (void)stamps;
  // C layout: col (lane & 15) -> kx*16 + c, row ((lane >> 4) * 4 + reg) -> co
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        wg_emit((ws ? ws + (size_t)blockIdx.x * (64 * 256) : dw) + (size_t)(f * 16 + grp * 4 + reg) * 256 + ky * 64 + kx * 16 + i16, ws != nullptr, acc[f][kx][reg]);
}

// deterministic mode, second stage: dw[i] += ws[0][i] + ws[1][i] + ... in split order (one thread per 4 elements)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long n4, int n_splits,
                                                           long long ws_stride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<const float4*>(ws)[i];
  for (int s = 1; s < n_splits; ++s) {
    const float4 b = reinterpret_cast<const float4*>(ws + (size_t)s * (size_t)ws_stride)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float4 d = reinterpret_cast<float4*>(dw)[i];
  d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
  reinterpret_cast<float4*>(dw)[i] = d;
}

int launch_wgrad_reduce(const WgradParams& p, int n_splits, long long n_elems, hipStream_t stream) {
  const long long n4 = n_elems / 4;
  vdqn_prof_begin("wgrad_reduce", 0.0, 4.0 * (double)n_elems * (n_splits + 2), stream);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, p.ws, p.dw, n4, n_splits, p.ws_stride);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

int launch_stem_wgrad(const WgradParams& p, hipStream_t stream) {
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&stem_wgrad_kernel), (size_t)kSwSmem);
  const int total_rows = p.n_img * 112;
  const int grid = total_rows < 512 ? total_rows : 512;
  const int rpb = (total_rows + grid - 1) / grid;
  vdqn_prof_begin("wgrad_stem<bf16>", 2.0 * p.M * 64 * 147, 2.0 * ((double)p.M * 64 + (double)p.n_img * 115 * 115 * 16) + 4.0 * 64 * 256, stream);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3((total_rows + rpb - 1) / rpb), dim3(256), kSwSmem, stream, (const bf16raw*)p.gy, (const bf16raw*)p.x, p.dw, total_rows,
                     rpb, p.gy_bytes, p.x_bytes, p.ws);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  if (p.ws) return launch_wgrad_reduce(p, (total_rows + rpb - 1) / rpb, 64 * 256, stream);
  return VDQN_OK;
}

int launch_wgrad_win(const WgradParams& p, int tiles, int splitk, hipStream_t stream) {
  constexpr int BCO = 64, BCI = 64, KP = 128, WR = KP + 8;
  const size_t smem_stage = (size_t)2 * KP * BCO * 2 + (size_t)(2 * WR) * BCI * 2 + kWgZeroBytes + kWgCodeBytes;  // gy tiles, x windows, zeros, border codes
  const size_t smem_epi = (size_t)2 * 64 * 64 * 4;
  const size_t smem = smem_stage > smem_epi ? smem_stage : smem_epi;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&wgrad_win_kernel), (size_t)smem);
  vdqn_prof_begin("wgrad_win<bf16,64>", 2.0 * p.M * p.co * p.taps * p.ci,
                  2.0 * ((double)p.M * p.ldg + (double)p.n_img * p.hi * p.wi * p.ci) + 4.0 * p.co * p.taps * p.ci, stream);
  hipLaunchKernelGGL(wgrad_win_kernel, dim3(tiles * splitk), dim3(256), smem, stream, p);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

// Which kernel a call runs on and how its pixel range is split (shared by the launch and by the workspace-size query).
struct WgradPlan {
  int variant;      // 0 stem kernel, 1 window 64x64, 3 generic
  int bt, ci_tiles, tiles, splitk, kchunk;
  int copies;       // partial copies of dw the deterministic mode stores (active splits, or blocks of the stem kernel)
  long long copy_elems;  // floats per copy
};

int g_wgrad_win_blocks_override = -1;  // tools/ab_inproc.py: vdqn_debug_set_wgrad_win_blocks

int plan_wgrad(const vdqn_wgrad_args* a, WgradPlan* pl) {
  VDQN_CHECK(a != nullptr, "vdqn_conv2d_wgrad: null args");
  VDQN_CHECK(a->dtype == VDQN_F32 || a->dtype == VDQN_BF16, "vdqn_conv2d_wgrad: bad dtype %d", a->dtype);
  VDQN_CHECK(a->ci % 64 == 0 && a->ldg % 64 == 0, "vdqn_conv2d_wgrad: ci=%d and ldg=%d must be multiples of 64", a->ci, a->ldg);
  VDQN_CHECK(a->stride == 1 || a->stride == 2, "vdqn_conv2d_wgrad: stride %d unsupported", a->stride);
  const int64_t M64 = (int64_t)a->n_img * a->ho * a->wo;
  VDQN_CHECK(M64 > 0 && M64 < (1 << 24), "vdqn_conv2d_wgrad: %lld output pixels out of range (< 2^24: split the batch)", (long long)M64);
  VDQN_CHECK(a->ho * a->wo < 65536, "vdqn_conv2d_wgrad: ho*wo too large");
  const int co_pad = (a->co + 63) / 64 * 64;
  VDQN_CHECK(a->ldg >= co_pad, "vdqn_conv2d_wgrad: gy rows (ldg=%d) must hold co padded to 64 (%d)", a->ldg, co_pad);
  const int M = (int)M64, taps = a->r * a->s;
  pl->copy_elems = (long long)co_pad * taps * a->ci;
  // 1 x 1 layers (the ResNet downsamples): with ONE tap the weight matrix is small, 128 x 128 tiles mean 2-8 tiles times ~200 splits,
  // and the 512 blocks' f32 atomics (blocks x tile elements) outweigh the arithmetic: 64 x 64 tiles emit a quarter of them
  static const int ds64 = [] { const char* e = getenv("VDQN_WGRAD_DS64"); return e ? atoi(e) : 1; }();
  const int bt = (co_pad % 128 == 0 && a->ci % 128 == 0 && !(ds64 && taps == 1 && a->dtype == VDQN_BF16)) ? 128 : 64;
  pl->bt = bt;
  pl->ci_tiles = a->ci / bt;
  pl->tiles = (co_pad / bt) * taps * pl->ci_tiles;
  int splitk = a->splitk;
  const int max_split = (M + 255) / 256;  // at least 8 K-steps per block
  if (splitk <= 0) {
    // 512 blocks = ONE round at 2 blocks per CU: measured 650 vs 555 TFLOP/s against two rounds of half-length blocks
    // (half the f32 atomics, twice the K-steps per prologue/epilogue); fewer than 512 leaves CUs idle (384: 540)
    static const int target = [] { const char* e = getenv("VDQN_WGRAD_BLOCKS"); return e ? atoi(e) : 512; }();
    splitk = target / pl->tiles;
    if (splitk > max_split) splitk = max_split;
    if (splitk < 1) splitk = 1;
  }
  pl->variant = 3;
  // VDQN_WGRAD_WINDOW=0: the 3x3 / stride-1 layers on the generic kernel too (A/B).  (Since round 3 the 64 x 64 window tiles take
  // every such layer: with the K loop's vector instructions cut they win on layer4 too — 0.91 + 0.13 against 0.70 + 0.40 ms per
  // update for window + generic launches, profiles/r03f_ab_wgrad_valu_diet.txt)
  static const int use_win = [] { const char* e = getenv("VDQN_WGRAD_WINDOW"); return e ? atoi(e) : 1; }();
  static const int use_stem = [] { const char* e = getenv("VDQN_WGRAD_STEM"); return e ? atoi(e) : 1; }();
  // window kernel: one block per (co tile, kernel ROW, ci tile) computes the three horizontal taps
  if (use_stem && a->dtype == VDQN_BF16 && a->r == 4 && a->s == 1 && a->ci == 64 && a->pix_stride == 16 && a->hi == 115 && a->wi == 115 && a->ho == 112 &&
      a->wo == 112 && a->co == 64 && a->ldg == 64 && a->stride == 1 && a->pad == 0 && a->splitk <= 0) {
    pl->variant = 0;
    const int total_rows = a->n_img * 112;
    const int grid = total_rows < 512 ? total_rows : 512;
    const int rpb = (total_rows + grid - 1) / grid;
    pl->copies = (total_rows + rpb - 1) / rpb;
    pl->copy_elems = 64 * 256;
    pl->splitk = 1;
    pl->kchunk = M;
    return VDQN_OK;
  }
  const bool win_geom = a->dtype == VDQN_BF16 && a->r == 3 && a->s == 3 && a->stride == 1 && a->pad == 1 && a->pix_stride == a->ci && a->wo >= 2 &&
                        a->hi == a->ho && a->wi == a->wo && a->ho * a->wo <= kWgCodeBytes;  // (border-code table of one image in LDS)
  if (use_win && win_geom) {
    pl->variant = 1;
    pl->ci_tiles = a->ci / 64;
    pl->tiles = (co_pad / 64) * 3 * pl->ci_tiles;
    // VDQN_WGRAD_WIN_BLOCKS: workgroups per window weight-gradient launch.  512 = one full round at two per CU, the fastest launch on
    // its own (0.903 ms per update for the 13 launches; 384: 0.938, 768: 1.090) — but with the weight gradients alternating between two
    // low-priority streams beside the data-gradient chain, slightly fewer and longer blocks give the shorter UPDATE: 448: 5.636-5.643,
    // 384: 5.648-5.655, 512: 5.653-5.671, 768: 5.746, 256: 5.703 ms (steady box, profiles/r04ab_ab_wgrad_win_blocks.txt)
    static const int target_env = [] { const char* e = getenv("VDQN_WGRAD_WIN_BLOCKS"); return e ? atoi(e) : 448; }();
    const int target_w = g_wgrad_win_blocks_override > 0 ? g_wgrad_win_blocks_override : target_env;
    splitk = a->splitk > 0 ? a->splitk : target_w / pl->tiles;
    if (splitk > max_split) splitk = max_split;
    if (splitk < 1) splitk = 1;
  }
  pl->splitk = splitk;
  pl->kchunk = ((M + splitk - 1) / splitk + 127) / 128 * 128;  // multiple of every kernel variant's K-step
  pl->copies = (M + pl->kchunk - 1) / pl->kchunk;              // splits with an empty pixel range store nothing
  return VDQN_OK;
}

}  // namespace

extern "C" void vdqn_debug_set_wgrad_win_blocks(int v) { g_wgrad_win_blocks_override = v; }  // measurement hook (not part of include/vdqn.h): -1 = VDQN_WGRAD_WIN_BLOCKS

extern "C" int64_t vdqn_conv2d_wgrad_workspace_bytes(const vdqn_wgrad_args* a) {
  WgradPlan pl;
  if (plan_wgrad(a, &pl) != VDQN_OK) return -1;
  return (int64_t)pl.copies * pl.copy_elems * 4;
}

extern "C" int vdqn_conv2d_wgrad(const vdqn_wgrad_args* a, void* stream) {
  WgradPlan pl;
  const int prc = plan_wgrad(a, &pl);
  if (prc != VDQN_OK) return prc;
  VDQN_CHECK(a->gy && a->x && a->dw, "vdqn_conv2d_wgrad: null tensor");
  const int64_t M64 = (int64_t)a->n_img * a->ho * a->wo;
  const int co_pad = (a->co + 63) / 64 * 64;
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  p.gy = a->gy; p.x = a->x; p.dw = a->dw;
  p.n_img = a->n_img; p.hi = a->hi; p.wi = a->wi; p.ci = a->ci; p.pix_stride = a->pix_stride;
  p.ho = a->ho; p.wo = a->wo; p.co = a->co; p.ldg = a->ldg; p.r = a->r; p.s = a->s; p.stride = a->stride; p.pad = a->pad;
  p.M = (int)M64;
  p.taps = a->r * a->s;
  const long long gyb = M64 * a->ldg * (a->dtype == VDQN_BF16 ? 2 : 4);
  const long long xb = (long long)a->n_img * a->hi * a->wi * a->pix_stride * (a->dtype == VDQN_BF16 ? 2 : 4);
  VDQN_CHECK(gyb < 0x7fffffffLL && xb < 0x7fffffffLL, "vdqn_conv2d_wgrad: operand larger than 2 GiB (split the batch)");
  VDQN_CHECK(((uintptr_t)a->gy & 15) == 0 && ((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->dw & 15) == 0, "vdqn_conv2d_wgrad: operands must be 16-byte aligned");
  p.gy_bytes = (int)gyb;
  p.x_bytes = (int)xb;
  p.d_howo = make_fastdiv((uint32_t)(a->ho * a->wo));
  p.d_wo = make_fastdiv((uint32_t)a->wo);
  p.ci_tiles = pl.ci_tiles;
  p.splitk = pl.splitk;
  p.kchunk = pl.kchunk;
  // deterministic mode: the caller's workspace takes one partial copy of dw per split, summed in split order afterwards
  p.ws = nullptr;
  p.ws_stride = pl.copy_elems;
  p.stamps = nullptr;
#ifdef VDQN_STAMP
  { extern void* g_stamp_buffer; p.stamps = g_stamp_buffer; }
#endif
  if (a->workspace) {
    VDQN_CHECK(a->workspace_bytes >= (int64_t)pl.copies * pl.copy_elems * 4, "vdqn_conv2d_wgrad: workspace of %lld bytes, %lld needed",
               (long long)a->workspace_bytes, (long long)pl.copies * pl.copy_elems * 4);
    VDQN_CHECK(((uintptr_t)a->workspace & 15) == 0, "vdqn_conv2d_wgrad: workspace must be 16-byte aligned");
    p.ws = reinterpret_cast<float*>(a->workspace);
  }
  int rc;
  if (pl.variant == 0) rc = launch_stem_wgrad(p, st);
  else if (pl.variant == 1) rc = launch_wgrad_win(p, pl.tiles, pl.splitk, st);
  else if (a->dtype == VDQN_BF16) rc = pl.bt == 128 ? launch_wgrad<bf16raw, 128>(p, pl.tiles, pl.splitk, st) : launch_wgrad<bf16raw, 64>(p, pl.tiles, pl.splitk, st);
  else rc = pl.bt == 128 ? launch_wgrad<float, 128>(p, pl.tiles, pl.splitk, st) : launch_wgrad<float, 64>(p, pl.tiles, pl.splitk, st);
  if (rc != VDQN_OK) return rc;
  if (p.ws && pl.variant != 0) {
    // only the rows below co: the kernels store nothing for the padding rows co .. co_pad - 1 of a copy (dw keeps its zeros there,
    // as in the atomic mode), and the caller's workspace is not initialised
    rc = launch_wgrad_reduce(p, pl.copies, (long long)a->co * p.taps * a->ci, st);
    if (rc != VDQN_OK) return rc;
  }
  if (a->dbias) {
    const int e16 = a->dtype == VDQN_BF16 ? 8 : 4;
    VDQN_CHECK(co_pad / e16 <= 256, "vdqn_conv2d_wgrad: dbias path supports up to %d channels", 256 * e16);
    int blocks = (p.M + 511) / 512;
    if (blocks > 256) blocks = 256;
    if (a->workspace) blocks = 1;  // deterministic mode: one block sums every row (its atomics then never race)
    const int rpb = (p.M + blocks - 1) / blocks;
    vdqn_prof_begin("colsum", 0.0, (double)p.M * co_pad * (a->dtype == VDQN_BF16 ? 2 : 4), st);
    if (a->dtype == VDQN_BF16)
      hipLaunchKernelGGL((colsum_kernel<bf16raw>), dim3(blocks), dim3(256), 0, st, (const bf16raw*)a->gy, a->dbias, p.M, co_pad, a->ldg, rpb);
    else
      hipLaunchKernelGGL((colsum_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)a->gy, a->dbias, p.M, co_pad, a->ldg, rpb);
    vdqn_prof_end(st);
    VDQN_LAUNCH_CHECK();
  }
  return VDQN_OK;
}

namespace {
// blocks of stem_wgrad_pool_kernel: whole pooled-row ranges of ONE image each, about 512 blocks in all
void stem_pool_grid(int n_img, int* pairs_per_block, int* blocks_per_img) {
  int bpi = 512 / n_img;
  bpi = bpi < 1 ? 1 : (bpi > 56 ? 56 : bpi);
  *pairs_per_block = (56 + bpi - 1) / bpi;
  *blocks_per_img = (56 + *pairs_per_block - 1) / *pairs_per_block;
}
}  // namespace

extern "C" int64_t vdqn_stem_wgrad_pool_workspace_bytes(int32_t n_img) {
  if (n_img <= 0) return 0;
  int ppb, bpi;
  stem_pool_grid(n_img, &ppb, &bpi);
  return (int64_t)n_img * bpi * 64 * 256 * 4;
}

extern "C" int vdqn_stem_wgrad_pool(const void* g_pool, const uint8_t* idx, const void* t_in, float* dw, int32_t n_img, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  VDQN_CHECK(g_pool && idx && t_in && dw && n_img > 0, "vdqn_stem_wgrad_pool: bad args");
  const int64_t gp_bytes = (int64_t)n_img * 56 * 56 * 64 * 2, idx_bytes = (int64_t)n_img * 56 * 56 * 64, x_bytes = (int64_t)n_img * 115 * 115 * 16 * 2;
  VDQN_CHECK(x_bytes < (1ll << 31) && gp_bytes < (1ll << 31), "vdqn_stem_wgrad_pool: %d images exceed the 2 GiB operand range (split the batch)", n_img);
  int ppb, bpi;
  stem_pool_grid(n_img, &ppb, &bpi);
  const int grid = n_img * bpi;
  float* ws = nullptr;
  if (workspace) {
    VDQN_CHECK(workspace_bytes >= vdqn_stem_wgrad_pool_workspace_bytes(n_img), "vdqn_stem_wgrad_pool: workspace of %lld bytes, need %lld",
               (long long)workspace_bytes, (long long)vdqn_stem_wgrad_pool_workspace_bytes(n_img));
    ws = reinterpret_cast<float*>(workspace);
  }
  hipStream_t st = (hipStream_t)stream;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&stem_wgrad_pool_kernel), (size_t)kSpSmem);
  const double M = (double)n_img * 112 * 112;
  vdqn_prof_begin("wgrad_stem_pool<bf16>", 2.0 * M * 64 * 147, (double)gp_bytes + (double)idx_bytes + (double)x_bytes + 4.0 * 64 * 256, st);
  void* stamps = nullptr;
#ifdef VDQN_STAMP
  { extern void* g_stamp_buffer; stamps = g_stamp_buffer; }
#endif
  hipLaunchKernelGGL(stem_wgrad_pool_kernel, dim3(grid), dim3(256), kSpSmem, st, (const bf16raw*)g_pool, idx, (const bf16raw*)t_in, dw, ppb, bpi, (int)gp_bytes,
                     (int)idx_bytes, (int)x_bytes, ws, stamps);
  vdqn_prof_end(st);
  VDQN_LAUNCH_CHECK();
  if (ws) {
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.ws = ws;
    p.dw = dw;
    p.ws_stride = 64 * 256;
    return launch_wgrad_reduce(p, grid, 64 * 256, st);
  }
  return VDQN_OK;
}
