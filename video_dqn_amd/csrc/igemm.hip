// Implicit-GEMM convolution / linear kernel for gfx950 (forward and data-gradient).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),   m = (img, ho, wo),  k = (r, s, c)
//
// * A is gathered on the fly from the NHWC activation tensor: one K-step is 128 contiguous bytes of one
//   (r, s) tap of one input pixel (64 bf16 / 32 f32 channels).  Both operands go HBM -> LDS directly with
//   `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB = 8 tile rows per wave-instruction): no staging VGPRs, no
//   ds_write.  Padding taps and rows past M need no branch: their lane offset is set out of the buffer
//   descriptor's range and the hardware writes zeros into LDS (probed: tools/probes/lds_dma_oob.hip).
// * Tiles: 128 (M) x BN (N, 64 or 128) x 128 bytes (K); 256 threads = 4 waves in a 2x2 grid, each wave owns
//   64 x BN/2 as 16x16 MFMA fragments (v_mfma_f32_16x16x32_bf16, or v_mfma_f32_16x16x4_f32 in the exact
//   f32 parity mode — same staging code, only the MFMA differs).
// * LDS: rows of 128 B; the DMA image is lane-linear, so the bank-conflict swizzle is applied on the SOURCE:
//   lane (row, p) fetches logical chunk p ^ (row & 7) and fragment reads XOR the same term.  Two buffers;
//   the DMA of step k+1 is in flight while step k's MFMAs run; one barrier per K-step.
// * Epilogue: accumulators -> LDS (f32) -> 16-byte vector loads of bias/residual/mask and 16-byte stores
//   (whole 128/256-byte output rows per 8/16 lanes).
// * blockIdx is remapped so the N-tiles of one M-tile run on the same XCD (A rows stay in that XCD's L2).
//
// Reference call sites this serves: the torch conv2d/linear (+BatchNorm eval, ReLU, residual) launched
// from archs/HabitatDQNMultiAction.py:30-31,49-53 and their backward (train_q_network.py:226).
#include <stdlib.h>

#include "common.h"

#ifndef VDQN_IGEMM_STAGES
#define VDQN_IGEMM_STAGES 2
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
  const void* in;
  const void* wt;
  const float* bias;
  const void* resid;
  const void* mask;
  void* out;
  float* out_f32;
  float* colsum_part;
  int n_img, hi, wi, ci, pix_stride, ho, wo, co, ldo, r, s, stride, pad, relu;
  int M, howo, ktot, nk, tiles_m, tiles_n;
  int cls_tile0[5], cls_h[2], cls_w[2];  // MODE 2: first tile of each output-parity class; class heights / widths
  long long in_bytes;
  int wt_bytes;
  int vec_ok;
};

constexpr unsigned kOob = 0x80000000u;  // voffset beyond any descriptor range (num_records <= 0x7fffffff)

// MODE 0: forward gather (h = oh*stride - pad + kr); 1: dgrad, stride 1 (h = oh + pad - kr);
//      2: dgrad, stride 2 (h = (oh + pad - kr) / 2 when even)
template <typename T, int BM, int BN, int MODE, int NSTAGE>
__global__ __launch_bounds__(2 * BM, NSTAGE == 3 && BN == 128 ? 1 : 2) void igemm_kernel(const IgemmParams p) {
  constexpr int NT = 2 * BM;    // threads: BM/64 x 2 waves, each owning 64 x BN/2 (BM = 256 is used for the 64-column
                                // layers: 16 waves per CU and half the weight traffic per MFMA)
  constexpr int RPS = NT / 8;   // tile rows staged per pass (8 lanes x 16 B per 128-byte row)
  constexpr int ESZ = (int)sizeof(T);
  constexpr int KC = 128 / ESZ;
  constexpr int NF = BN / 32;
  constexpr int BROWS = BN / RPS;
  constexpr int PSTR = RPS * 128;  // LDS byte distance between a thread's consecutive DMA pieces
  constexpr int LDC = BN + 4;  // f32 row stride of the epilogue tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + NSTAGE * BM * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lb % (uint32_t)p.tiles_n), tile_m = (int)(lb / (uint32_t)p.tiles_n);
  const int n0 = tile_n * BN;
  // MODE 2 (stride-2 data gradient): an output pixel (oh, ow) only receives the taps with kr = oh + pad (mod 2) and
  // ks = ow + pad (mod 2).  Rows are therefore enumerated parity class by parity class ((oh & 1, ow & 1): 4 classes
  // of Ho/2 x Wo/2 pixels, tiles never straddle classes) and each tile walks only ITS taps: 9 tap-classes in total
  // instead of 4 x 9 with three quarters of the A rows zero.
  int m0 = tile_m * BM, nk = p.nk, kr0 = 0, ks0 = 0, cls_ph = 0, cls_pw = 0;
  if constexpr (MODE == 2) {
    const int cls = (tile_m >= p.cls_tile0[1]) + (tile_m >= p.cls_tile0[2]) + (tile_m >= p.cls_tile0[3]);
    m0 = (tile_m - p.cls_tile0[cls]) * BM;  // first row inside the class
    cls_ph = cls >> 1;
    cls_pw = cls & 1;
    kr0 = (cls_ph + p.pad) & 1;
    ks0 = (cls_pw + p.pad) & 1;
    const int nkr = p.r > kr0 ? (p.r - kr0 + 1) / 2 : 0, nks = p.s > ks0 ? (p.s - ks0 + 1) / 2 : 0;
    nk = nkr * nks * (p.ci / KC);
  }
  const int row_w = MODE == 2 ? p.cls_w[cls_pw] : p.wo;
  const int pix_per_img = MODE == 2 ? p.cls_h[cls_ph] * row_w : p.howo;
  const int rows_total = MODE == 2 ? p.n_img * pix_per_img : p.M;
  const int lrow = tid >> 3;                        // tile row this thread stages (+32 i)
  const int lchunk = (tid & 7) ^ (lrow & 7);        // logical 16-byte chunk it fetches (source-side swizzle)

  // ---- buffer descriptors (4 SGPRs each): A relative to the first image of this tile, B = whole weight tensor.
  // The DMA is issued from inline asm: hipcc would otherwise wait vmcnt(0) before the first ds_read that follows
  // a pending LDS-DMA (it cannot prove the buffers distinct), which serialises the copy behind the MFMAs.
  const int img0 = m0 / pix_per_img;
  const long long img_bytes = (long long)p.hi * p.wi * p.pix_stride * ESZ;
  const long long a_base_off = (long long)img0 * img_bytes;
  long long a_rem = p.in_bytes - a_base_off;
  if (a_rem > 0x7fffffffLL) a_rem = 0x7fffffffLL;
  const unsigned long long a_ptr = (unsigned long long)((const unsigned char*)p.in + a_base_off);
  const unsigned long long b_ptr = (unsigned long long)p.wt;
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)a_rem), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};

  // ---- per-row gather state (4 A rows per thread) ----
  uint32_t a_off[4];
  int a_hb[4], a_wb[4];
  const int pixB = p.pix_stride * ESZ;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lrow + RPS * i;
    const bool ok = m < rows_total;
    const int mm = ok ? m : m0;
    const int img = mm / pix_per_img;
    const int rem = mm - img * pix_per_img;
    int oh = rem / row_w;
    int ow = rem - oh * row_w;
    if constexpr (MODE == 2) {
      oh = 2 * oh + cls_ph;
      ow = 2 * ow + cls_pw;
    }
    int hb, wb, hq, wq;
    if (MODE == 0) {
      hb = oh * p.stride - p.pad;
      wb = ow * p.stride - p.pad;
      hq = hb;
      wq = wb;
    } else {
      hb = oh + p.pad;
      wb = ow + p.pad;
      hq = MODE == 2 ? (hb >> 1) : hb;
      wq = MODE == 2 ? (wb >> 1) : wb;
    }
    a_off[i] = (uint32_t)(((img - img0) * p.hi + hq) * p.wi + wq) * (uint32_t)pixB + (uint32_t)(lchunk * 16);
    a_hb[i] = ok ? hb : -(1 << 20);
    a_wb[i] = wb;
  }
  uint32_t b_off[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_off[i] = (uint32_t)(n0 + lrow + RPS * i) * (uint32_t)(p.ktot * ESZ) + (uint32_t)(lchunk * 16);

  // LDS byte addresses (wave-uniform) of this wave's DMA pieces: piece i of an operand covers tile rows 32 i + 8 wave .. +7
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t lds_wave = lds_base + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * (8 * 128);

#define VDQN_ISSUE(BUF, KR, KS, C0, KSTEP)                                                                          \
  {                                                                                                                 \
    const int delta_ = (MODE == 0   ? (((KR)*p.wi + (KS)) * p.pix_stride + (C0))                                    \
                        : MODE == 1 ? ((C0) - ((KR)*p.wi + (KS)) * p.pix_stride)                                    \
                                    : ((C0) - (((KR) >> 1) * p.wi + ((KS) >> 1)) * p.pix_stride)) *                 \
                       ESZ;                                                                                         \
    uint32_t vo_[4];                                                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                              \
      bool ok_;                                                                                                     \
      if (MODE == 0) {                                                                                              \
        ok_ = ((unsigned)(a_hb[i_] + (KR)) < (unsigned)p.hi) && ((unsigned)(a_wb[i_] + (KS)) < (unsigned)p.wi);     \
      } else if (MODE == 1) {                                                                                       \
        ok_ = ((unsigned)(a_hb[i_] - (KR)) < (unsigned)p.hi) && ((unsigned)(a_wb[i_] - (KS)) < (unsigned)p.wi);     \
      } else {                                                                                                      \
        const int th_ = a_hb[i_] - (KR), tw_ = a_wb[i_] - (KS);                                                     \
        ok_ = ((unsigned)(th_ >> 1) < (unsigned)p.hi) && ((unsigned)(tw_ >> 1) < (unsigned)p.wi); /* parity holds by class */ \
      }                                                                                                             \
      vo_[i_] = ok_ ? a_off[i_] + (uint32_t)delta_ : kOob;                                                          \
    }                                                                                                               \
    const uint32_t la_ = lds_wave + (uint32_t)(BUF) * (BM * 128);                                                   \
    asm volatile(                                                                                                   \
        "s_nop 4\n\t"                                                                                               \
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %5, 0 offen lds\n\t"                               \
        "s_add_u32 m0, %4, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, 0 offen lds\n\t"                           \
        "s_add_u32 m0, %4, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, 0 offen lds\n\t"                           \
        "s_add_u32 m0, %4, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, 0 offen lds"                                \
        ::"v"(vo_[0]), "v"(vo_[1]), "v"(vo_[2]), "v"(vo_[3]), "s"(la_), "s"(rs_a), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
        : "memory", "scc");                                                                                         \
    const uint32_t lb_ = lds_wave + (uint32_t)(NSTAGE * BM * 128) + (uint32_t)(BUF) * (BN * 128);                   \
    const int so_ = (KSTEP)*128;                                                                                    \
    if constexpr (BROWS == 4) {                                                                                     \
      asm volatile(                                                                                                 \
          "s_nop 4\n\t"                                                                                             \
          "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %5, %6 offen lds\n\t"                            \
          "s_add_u32 m0, %4, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"                        \
          "s_add_u32 m0, %4, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"                        \
          "s_add_u32 m0, %4, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds"                             \
          ::"v"(b_off[0]), "v"(b_off[BROWS > 1 ? 1 : 0]), "v"(b_off[BROWS > 2 ? 2 : 0]), "v"(b_off[BROWS > 2 ? 3 : 0]), "s"(lb_),    \
          "s"(rs_b), "s"(so_), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR)                                              \
          : "memory", "scc");                                                                                       \
    } else if constexpr (BROWS == 2) {                                                                              \
      asm volatile(                                                                                                 \
          "s_nop 4\n\t"                                                                                             \
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %4 offen lds\n\t"                            \
          "s_add_u32 m0, %2, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds"                             \
          ::"v"(b_off[0]), "v"(b_off[BROWS > 1 ? 1 : 0]), "s"(lb_), "s"(rs_b), "s"(so_), "n"(PSTR)                  \
          : "memory", "scc");                                                                                       \
    } else {                                                                                                        \
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"             \
                   ::"v"(b_off[0]), "s"(lb_), "s"(rs_b), "s"(so_)                                                   \
                   : "memory");                                                                                     \
    }                                                                                                               \
  }
#define VDQN_ADVANCE()                 \
  {                                    \
    ic0 += KC;                         \
    if (ic0 >= p.ci) {                 \
      ic0 = 0;                         \
      iks += (MODE == 2 ? 2 : 1);      \
      if (iks >= p.s) {                \
        iks = ks0;                     \
        ikr += (MODE == 2 ? 2 : 1);    \
      }                                \
    }                                  \
  }
  // index of a K-step inside a weight row ([r][s][ci], 128-byte steps)
#define VDQN_WSTEP() (MODE == 2 ? ((ikr * p.s + iks) * (p.ci / KC) + ic0 / KC) : issued)

  f32x4 acc[4][NF];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, g = lane >> 4;
  const int sw = i16 & 7;

  // ---- main loop: K-steps enumerate (r, s, c0) with c0 fastest; NSTAGE LDS buffers, NSTAGE-1 tiles in flight ----
  int ikr = kr0, iks = ks0, ic0 = 0;  // coordinates of the next K-step to issue
  int issued = 0;
#pragma unroll
  for (int s_ = 0; s_ < NSTAGE - 1; ++s_) {
    if (issued < nk) {
      VDQN_ISSUE(s_, ikr, iks, ic0, VDQN_WSTEP())
      VDQN_ADVANCE()
      ++issued;
    }
  }
  int buf = 0, ibuf = NSTAGE - 1;
  for (int k = 0; k < nk; ++k) {
    // tile k has landed once at most (issued - k - 1) younger tiles are still outstanding
    if (NSTAGE == 3 && issued - k - 1 >= 1) {
      if constexpr (BROWS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (BROWS == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // every wave's pieces of tile k are visible; every wave is done with tile k-1
    if (issued < nk) {
      VDQN_ISSUE(ibuf, ikr, iks, ic0, VDQN_WSTEP())
      VDQN_ADVANCE()
      ++issued;
    }
    const unsigned char* a = sA + buf * (BM * 128) + (wr * 64 + i16) * 128;
    const unsigned char* b = sB + buf * (BN * 128) + (wc * (BN / 2) + i16) * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int coff = (((g + 4 * h) ^ sw) << 4);
      uint4 af[4], bfr[NF];
#pragma unroll
      for (int f = 0; f < 4; ++f) af[f] = *reinterpret_cast<const uint4*>(a + f * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < NF; ++j) bfr[j] = *reinterpret_cast<const uint4*>(b + j * 16 * 128 + coff);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[f]), __builtin_bit_cast(bf16x8, bfr[j]),
                                                                acc[f][j], 0, 0, 0);
          } else {
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].x), __uint_as_float(bfr[j].x), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].y), __uint_as_float(bfr[j].y), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].z), __uint_as_float(bfr[j].z), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].w), __uint_as_float(bfr[j].w), acc[f][j], 0, 0, 0);
          }
        }
    }
    buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    ibuf = (ibuf + 1 == NSTAGE) ? 0 : ibuf + 1;
  }
  __syncthreads();  // all waves done reading LDS before the epilogue reuses it
#undef VDQN_ISSUE
#undef VDQN_ADVANCE
#undef VDQN_WSTEP

  // ---- epilogue: accumulators -> LDS f32 tile (C layout: col = lane & 15, row = (lane >> 4) * 4 + reg) ----
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        sC[(wr * 64 + f * 16 + g * 4 + reg) * LDC + wc * (BN / 2) + j * 16 + i16] = acc[f][j][reg];
  __syncthreads();

  T* __restrict__ out = (T*)p.out;
  const T* __restrict__ resid = (const T*)p.resid;
  const T* __restrict__ mask = (const T*)p.mask;
  constexpr int TPR = BN / 8;         // threads per tile row (8 columns each)
  constexpr int RPP = NT / TPR;       // rows per pass
  const int col8 = (tid % TPR) * 8;
  const int n = n0 + col8;
  float cs[8];  // per-thread column sums of the values this tile stores (for the BN-shift / bias gradient)
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  if (n < p.co) {
    const bool vec = p.vec_ok && (n + 8 <= p.co);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (p.bias && n + e < p.co) ? p.bias[n + e] : 0.f;
#pragma unroll 2
    for (int r0 = tid / TPR; r0 < BM; r0 += RPP) {
      int m = m0 + r0;
      if (m >= rows_total) break;
      if constexpr (MODE == 2) {  // class-local row -> output pixel
        const int img = m / pix_per_img;
        const int rem = m - img * pix_per_img;
        const int ohc = rem / row_w;
        m = (img * p.ho + 2 * ohc + cls_ph) * p.wo + 2 * (rem - ohc * row_w) + cls_pw;
      }
      const size_t o = (size_t)m * p.ldo + n;
      float v[8];
      const float4 c0v = *reinterpret_cast<const float4*>(sC + r0 * LDC + col8);
      const float4 c1v = *reinterpret_cast<const float4*>(sC + r0 * LDC + col8 + 4);
      v[0] = c0v.x + bv[0]; v[1] = c0v.y + bv[1]; v[2] = c0v.z + bv[2]; v[3] = c0v.w + bv[3];
      v[4] = c1v.x + bv[4]; v[5] = c1v.y + bv[5]; v[6] = c1v.z + bv[6]; v[7] = c1v.w + bv[7];
      if (vec) {
        T rv[8], mv[8], ov[8];
        if (resid) {
          if constexpr (ESZ == 2) *reinterpret_cast<uint4*>(rv) = *reinterpret_cast<const uint4*>(resid + o);
          else { reinterpret_cast<uint4*>(rv)[0] = reinterpret_cast<const uint4*>(resid + o)[0]; reinterpret_cast<uint4*>(rv)[1] = reinterpret_cast<const uint4*>(resid + o)[1]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += to_f32<T>(rv[e]);
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mask) {
          if constexpr (ESZ == 2) *reinterpret_cast<uint4*>(mv) = *reinterpret_cast<const uint4*>(mask + o);
          else { reinterpret_cast<uint4*>(mv)[0] = reinterpret_cast<const uint4*>(mask + o)[0]; reinterpret_cast<uint4*>(mv)[1] = reinterpret_cast<const uint4*>(mask + o)[1]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (to_f32<T>(mv[e]) > 0.f) ? v[e] : 0.f;
        }
        if (out) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ov[e] = from_f32<T>(v[e]);
            cs[e] += to_f32<T>(ov[e]);
          }
          if constexpr (ESZ == 2) *reinterpret_cast<uint4*>(out + o) = *reinterpret_cast<const uint4*>(ov);
          else { reinterpret_cast<uint4*>(out + o)[0] = reinterpret_cast<const uint4*>(ov)[0]; reinterpret_cast<uint4*>(out + o)[1] = reinterpret_cast<const uint4*>(ov)[1]; }
        }
        if (p.out_f32) {
          *reinterpret_cast<float4*>(p.out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(p.out_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      } else {
        for (int e = 0; e < 8 && n + e < p.co; ++e) {
          float x = v[e];
          if (resid) x += to_f32<T>(resid[o + e]);
          if (p.relu) x = fmaxf(x, 0.f);
          if (mask) x = (to_f32<T>(mask[o + e]) > 0.f) ? x : 0.f;
          if (out) {
            out[o + e] = from_f32<T>(x);
            cs[e] += to_f32<T>(from_f32<T>(x));
          }
          if (p.out_f32) p.out_f32[o + e] = x;
        }
      }
    }
  }
  if (p.colsum_part) {  // uniform branch: partial column sums of this tile -> colsum_part[tile_m][ldo]
    __syncthreads();    // everyone has read the staging tile: reuse its head for the reduction
    float* sR = sC;
#pragma unroll
    for (int e = 0; e < 8; ++e) sR[(tid / TPR) * BN + col8 + e] = cs[e];
    __syncthreads();
    if (tid < BN && n0 + tid < p.co) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) t += sR[r * BN + tid];
      if constexpr (BM == 256) {  // consumers sum ceil(M/128) entries: this tile covers two of them
        p.colsum_part[(size_t)(2 * tile_m) * p.ldo + n0 + tid] = t;
        if ((2 * tile_m + 1) * 128 < p.M) p.colsum_part[(size_t)(2 * tile_m + 1) * p.ldo + n0 + tid] = 0.f;
      } else {
        p.colsum_part[(size_t)tile_m * p.ldo + n0 + tid] = t;
      }
    }
  }
}

template <typename T, int BM, int BN, int MODE, int NSTAGE = VDQN_IGEMM_STAGES>
int launch_igemm(const IgemmParams& p, hipStream_t stream) {
  const size_t main_bytes = NSTAGE * (BM + BN) * 128, epi_bytes = BM * (BN + 4) * 4;
  const size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, MODE, NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
  const double esz = sizeof(T);
  // one tag per kernel symbol (T, BM, BN, MODE), so bench.py rows line up with rocprofv3's kernel names
  static const char* const kTag[2][3][3] = {{{"igemm<bf16,64,fwd>", "igemm<bf16,64,dgrad>", "igemm<bf16,64,dgrad_s2>"},
                                             {"igemm<bf16,128,fwd>", "igemm<bf16,128,dgrad>", "igemm<bf16,128,dgrad_s2>"},
                                             {"igemm<bf16,256x64,fwd>", "igemm<bf16,256x64,dgrad>", "igemm<bf16,256x64,dgrad_s2>"}},
                                            {{"igemm<f32,64,fwd>", "igemm<f32,64,dgrad>", "igemm<f32,64,dgrad_s2>"},
                                             {"igemm<f32,128,fwd>", "igemm<f32,128,dgrad>", "igemm<f32,128,dgrad_s2>"},
                                             {"igemm<f32,256x64,fwd>", "igemm<f32,256x64,dgrad>", "igemm<f32,256x64,dgrad_s2>"}}};
  vdqn_prof_begin(kTag[sizeof(T) == 2 ? 0 : 1][BM == 256 ? 2 : (BN == 128 ? 1 : 0)][MODE],
                  2.0 * p.M * p.co * p.ktot,
                  esz * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))),
                  stream);
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, MODE, NSTAGE>), dim3(grid), dim3(2 * BM), smem, stream, p);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

template <typename T, int BM, int BN, int NSTAGE = VDQN_IGEMM_STAGES>
int launch_mode(const IgemmParams& p, int mode, hipStream_t st) {
  if (mode == 0) return launch_igemm<T, BM, BN, 0, NSTAGE>(p, st);
  if (mode == 1) return launch_igemm<T, BM, BN, 1, NSTAGE>(p, st);
  if constexpr (BM == 128) return launch_igemm<T, BM, BN, 2, NSTAGE>(p, st);
  return VDQN_ERR_INVALID;
}

}  // namespace

extern "C" int vdqn_conv2d(const vdqn_conv_args* a, void* stream) {
  VDQN_CHECK(a != nullptr, "vdqn_conv2d: null args");
  VDQN_CHECK(a->dtype == VDQN_F32 || a->dtype == VDQN_BF16, "vdqn_conv2d: bad dtype %d", a->dtype);
  const int esz = a->dtype == VDQN_BF16 ? 2 : 4;
  const int kc = 128 / esz;
  VDQN_CHECK(a->in && a->wt && (a->out || a->out_f32), "vdqn_conv2d: null tensor");
  VDQN_CHECK(a->ci > 0 && a->ci % kc == 0, "vdqn_conv2d: ci=%d must be a multiple of %d", a->ci, kc);
  VDQN_CHECK(a->stride == 1 || a->stride == 2, "vdqn_conv2d: stride %d unsupported", a->stride);
  VDQN_CHECK(a->mode == 0 || a->mode == 1, "vdqn_conv2d: bad mode %d", a->mode);
  VDQN_CHECK(a->n_img > 0 && a->ho > 0 && a->wo > 0 && a->co > 0 && a->ldo >= a->co, "vdqn_conv2d: bad dims");
  VDQN_CHECK((int64_t)a->n_img * a->ho * a->wo < (1ll << 31), "vdqn_conv2d: too many output pixels");
  VDQN_CHECK((a->pix_stride * esz) % 16 == 0 && (((uintptr_t)a->in | (uintptr_t)a->wt) & 15) == 0, "vdqn_conv2d: in/wt must be 16-byte aligned");
  IgemmParams p;
  p.in = a->in; p.wt = a->wt; p.bias = a->bias; p.resid = a->resid; p.mask = a->mask; p.out = a->out; p.out_f32 = a->out_f32; p.colsum_part = a->colsum_part;
  p.n_img = a->n_img; p.hi = a->hi; p.wi = a->wi; p.ci = a->ci; p.pix_stride = a->pix_stride;
  p.ho = a->ho; p.wo = a->wo; p.co = a->co; p.ldo = a->ldo; p.r = a->r; p.s = a->s; p.stride = a->stride; p.pad = a->pad;
  p.relu = a->relu;
  p.M = a->n_img * a->ho * a->wo;
  p.howo = a->ho * a->wo;
  p.ktot = a->r * a->s * a->ci;
  p.nk = p.ktot / kc;
  const int bn = (a->co % 128 == 0) ? 128 : 64;
  p.tiles_m = (p.M + 127) / 128;
  memset(p.cls_tile0, 0, sizeof(p.cls_tile0));
  p.cls_h[0] = p.cls_h[1] = p.cls_w[0] = p.cls_w[1] = 0;
  if (a->mode == 1 && a->stride == 2) {  // tiles are laid out parity class by parity class (see the kernel)
    p.cls_h[0] = (a->ho + 1) / 2; p.cls_h[1] = a->ho / 2;
    p.cls_w[0] = (a->wo + 1) / 2; p.cls_w[1] = a->wo / 2;
    for (int c = 0; c < 4; ++c) p.cls_tile0[c + 1] = p.cls_tile0[c] + (a->n_img * p.cls_h[c >> 1] * p.cls_w[c & 1] + 127) / 128;
    p.tiles_m = p.cls_tile0[4];
  }
  p.tiles_n = (a->co + bn - 1) / bn;
  p.in_bytes = (long long)a->n_img * a->hi * a->wi * a->pix_stride * esz;
  const long long wtb = (long long)p.tiles_n * bn * p.ktot * esz;
  VDQN_CHECK(wtb < 0x7fffffffLL, "vdqn_conv2d: weight tensor too large");
  p.wt_bytes = (int)wtb;
  // a 128-row tile spans at most 128 images; its gather window must stay below 2 GiB
  const long long img_bytes = (long long)a->hi * a->wi * a->pix_stride * esz;
  const long long span_imgs = 128 / (p.howo > 0 ? p.howo : 1) + 2;
  VDQN_CHECK(span_imgs * img_bytes < 0x7fffffffLL, "vdqn_conv2d: image too large for one tile's gather window");
  const uintptr_t al = (uintptr_t)a->out | (uintptr_t)a->out_f32 | (uintptr_t)a->resid | (uintptr_t)a->mask;
  p.vec_ok = ((a->ldo * esz) % 16 == 0) && ((al & 15) == 0) && (a->ldo % 8 == 0);
  const int mode = a->mode == 0 ? 0 : (a->stride == 2 ? 2 : 1);
  hipStream_t st = (hipStream_t)stream;
  // 64-column layers with many rows: 256-row tiles, 8 waves (more MFMA work per DMA round trip, half the weight traffic)
  // (VDQN_BM256_MIN_ROWS overrides the row threshold: tests lower it to reach this variant with small tensors, a huge value disables it)
  static const long long min256 = [] { const char* e = getenv("VDQN_BM256_MIN_ROWS"); return e ? atoll(e) : 256ll * 1024; }();
  if (bn == 64 && mode != 2 && p.M >= min256) {
    p.tiles_m = (p.M + 255) / 256;
    return a->dtype == VDQN_BF16 ? launch_mode<bf16raw, 256, 64>(p, mode, st) : launch_mode<float, 256, 64>(p, mode, st);
  }
  if (a->dtype == VDQN_BF16) return bn == 128 ? launch_mode<bf16raw, 128, 128>(p, mode, st) : launch_mode<bf16raw, 128, 64>(p, mode, st);
  return bn == 128 ? launch_mode<float, 128, 128>(p, mode, st) : launch_mode<float, 128, 64>(p, mode, st);
}
