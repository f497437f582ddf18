// Implicit-GEMM convolution / linear kernel for gfx950 (forward and data-gradient).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),   m = (img, ho, wo),  k = (r, s, c)
//
// * A is gathered on the fly from the NHWC activation tensor: one K-step is 128 contiguous bytes of one
//   (r, s) tap of one input pixel (64 bf16 / 32 f32 channels).  Both operands go HBM -> LDS directly with
//   `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB = 8 tile rows per wave-instruction): no staging VGPRs, no
//   ds_write.  Padding taps and rows past M need no branch: their lane offset is set out of the buffer
//   descriptor's range and the hardware writes zeros into LDS (probed: tools/probes/lds_dma_oob.hip).
// * Tiles: BM (128 | 256) x BN (64 | 128) x 128 bytes of K; a (BM/64) x WN grid of waves, each owning 64 x BN/WN as
//   16x16 MFMA fragments (v_mfma_f32_16x16x32_bf16, or v_mfma_f32_16x16x4_f32 in the exact f32 parity mode — same
//   staging code, only the MFMA differs).
// * LDS: rows of 128 B; the DMA image is lane-linear, so the bank-conflict swizzle is applied on the SOURCE:
//   lane (row, p) fetches logical chunk p ^ key(row) and fragment reads XOR the same term.  Two LDS buffers and two
//   register sets of fragments: the DMA of tile k+2 and the ds_reads of tile k+1 are both in flight underneath the
//   MFMAs of tile k; one barrier per K-step, no LDS wait in front of any MFMA.
// * The weights are the MFMA's FIRST operand, so the accumulator lane (i16, g) holds 4 consecutive output channels of
//   pixel i16; the weight fragment rows are read in a permuted order (channel g*4NF + j*4 + r from fragment j) so that a
//   lane's NF fragments are 4*NF CONSECUTIVE channels.  The epilogue therefore needs no LDS transpose: every lane adds
//   bias / residual, applies ReLU / mask and stores 16..64 contiguous bytes; 4 lanes cover a whole output row of the
//   wave's tile (full 128-byte lines), and all residual/mask loads of a lane are issued before the first is used.
// * blockIdx is remapped so the N-tiles of one M-tile run on the same XCD (A rows stay in that XCD's L2).
//
// Reference call sites this serves: the torch conv2d/linear (+BatchNorm eval, ReLU, residual) launched
// from archs/HabitatDQNMultiAction.py:30-31,49-53 and their backward (train_q_network.py:226).
#include <stdlib.h>

#include "common.h"

int vdqn_stem_bf16(const void* t_in, const void* wt, const float* bias, void* pool, void* idx, int n_img, int n_idx_img, hipStream_t st);  // stem.hip
int vdqn_launch_win9s(const void* igemm_params, hipStream_t stream);                                                              // win9s.hip
int vdqn_win9s_supports(int cpk, int has_sib);
int vdqn_launch_win9d(const void* igemm_params, hipStream_t stream);                                                              // win9d.hip
int vdqn_win9d_supports(int ci, int co, int has_sib, int ci2);                                                                    // win9d.hip                                                                                    // win9s.hip
int vdqn_launch_win9u(const void* igemm_params, int mode, hipStream_t stream);                                                    // win9.hip

#include "igemm_common.h"

namespace {

// MODE 0: forward gather (h = oh*stride - pad + kr); 1: dgrad, stride 1 (h = oh + pad - kr);
//      2: dgrad, stride 2 (h = (oh + pad - kr) / 2 when even)
//      3: the ResNet stem — conv1 (as a 4x4/1 conv on the space-to-depth operand) + folded BatchNorm + ReLU + the 3x3/2
//         max-pool in one kernel: a tile is a 16x16 patch of conv pixels (rows 14 ty - 1 .., cols 14 tx - 1 ..) that holds
//         every window of a 7x7 patch of pooled pixels, so the 112x112x64 conv output never goes to HBM (the patches
//         overlap by two rows/columns: 30 % more MFMA work on a kernel whose cost is its output traffic)
// WN = waves along N.  256 x 64 tiles with WN = 1 give the 64-output-channel layers the same 64x64 wave tile (16 MFMAs
// per 8 fragment reads) as the 128x128 kernel.
// Scheduling pattern of a K-step's block (bf16): one MFMA, one VALU, one LDS read, N times, then the remaining MFMAs.
// The fragment reads of the NEXT K-step and their address arithmetic then issue in the shadow of this step's MFMAs instead
// of in front of them: +4 % on the window kernel inside the step, +5-9 % per layer (experiments/README.md).
// NS = staging buffers per operand: 2 (one K-step of DMA lead; two workgroups per CU hide the rest) or 4 (three K-steps of
// lead, counted vmcnt waits) for launches with at most one tile per CU, where nothing else hides the ~0.85 us DMA round trip
// of every K-step (the head's skinny GEMMs: features.8, top.*).
template <typename T, int BM, int BN, int MODE, int WN, int NS = 2>
__global__ __launch_bounds__(BM * WN, 2) void igemm_kernel(const IgemmParams p) {
  constexpr int NT = BM * WN;   // threads: BM/64 x WN waves, each owning 64 x BN/WN
  constexpr int RPS = NT / 8;   // tile rows staged per pass (8 lanes x 16 B per 128-byte row)
  constexpr int ESZ = (int)sizeof(T);
  constexpr int KC = 128 / ESZ;
  constexpr int NF = BN / (16 * WN);  // weight fragments per wave
  constexpr int CPL = 4 * NF;         // consecutive output channels a lane ends up with
  constexpr int BROWS = BN / RPS;
  constexpr int AROWS = BM / RPS;  // A rows staged per thread (4 or 8)
  static_assert(AROWS == 4 || AROWS == 8, "A staging is issued in groups of four pieces");
  static_assert(NF == 2 || NF == 4, "wave tile is 64 x 32 or 64 x 64");
  constexpr int PSTR = RPS * 128;  // LDS byte distance between a thread's consecutive DMA pieces
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(NS == 2 || (NS == 4 && (MODE == 0 || MODE == 1)), "deep ring: forward / stride-1 data gradient only");
  unsigned char* sA = smem;
  unsigned char* sB = smem + NS * BM * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lb % (uint32_t)p.tiles_n), tile_m = (int)(lb / (uint32_t)p.tiles_n);
  // fused sibling 1x1 (forward): this tile belongs to the 1x1 convolution — its own column range, weights and epilogue; K walks the
  // channel chunks of the centre tap only
  const bool sib = MODE == 0 && tile_n >= p.tiles_n1;
  const int n0 = (sib ? tile_n - p.tiles_n1 : tile_n) * BN;
  // MODE 2 (stride-2 data gradient): an output pixel (oh, ow) only receives the taps with kr = oh + pad (mod 2) and
  // ks = ow + pad (mod 2).  Rows are therefore enumerated parity class by parity class ((oh & 1, ow & 1): 4 classes
  // of Ho/2 x Wo/2 pixels, tiles never straddle classes) and each tile walks only ITS taps: 9 tap-classes in total
  // instead of 4 x 9 with three quarters of the A rows zero.
  int m0 = tile_m * BM, nk = p.nk, kr0 = 0, ks0 = 0, cls_ph = 0, cls_pw = 0;
  if constexpr (MODE == 2) {
    // class-major tile order, or (all classes equally long: every even-sized image) the four classes of one row block in
    // consecutive tiles — they read the same gy rows, which the XCD-contiguous tile order then finds in that XCD's L2
    const int cls = p.cls_interleave ? (tile_m & 3) : (tile_m >= p.cls_tile0[1]) + (tile_m >= p.cls_tile0[2]) + (tile_m >= p.cls_tile0[3]);
    m0 = (p.cls_interleave ? (tile_m >> 2) : (tile_m - p.cls_tile0[cls])) * BM;  // first row inside the class
    cls_ph = cls >> 1;
    cls_pw = cls & 1;
    kr0 = (cls_ph + p.pad) & 1;
    ks0 = (cls_pw + p.pad) & 1;
    const int nkr = p.r > kr0 ? (p.r - kr0 + 1) / 2 : 0, nks = p.s > ks0 ? (p.s - ks0 + 1) / 2 : 0;
    nk = nkr * nks * (p.ci / KC);
  }
  if (sib) {
    kr0 = p.r >> 1;  // centre tap
    ks0 = p.s >> 1;
    nk = p.ci / KC;
  }
  // fused sibling (stride-2 data gradient): class (0, 0) tiles run extra K-steps over the sibling's output gradient
  const int nk2 = (MODE == 2 && p.in2 != nullptr && cls_ph == 0 && cls_pw == 0) ? p.ci2 / KC : 0;
  const int nk_all = nk + nk2;
  constexpr bool FWD = (MODE == 0 || MODE == 3);
  const int st_img = MODE == 3 ? tile_m >> 6 : 0, st_ty = (tile_m >> 3) & 7, st_tx = tile_m & 7;  // MODE 3 tile coordinates
  const int row_w = MODE == 2 ? p.cls_w[cls_pw] : p.wo;
  const int pix_per_img = MODE == 2 ? p.cls_h[cls_ph] * row_w : p.howo;
  const int rows_total = MODE == 2 ? p.n_img * pix_per_img : p.M;
  const int lrow = tid >> 3;  // tile row this thread stages (+RPS i)
  // logical 16-byte chunk it fetches (source-side swizzle).  A rows are read by lanes i16 = row & 15: key = row & 7.
  // Weight rows are read in the permuted order row(j, i16) = (i16 >> 2) * CPL + j * 4 + (i16 & 3); the key that is
  // again i16 & 7 for the reading lane is ((row / CPL) & 1) * 4 + (row & 3).
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int lchunk_b = (tid & 7) ^ ((((lrow / CPL) & 1) << 2) | (lrow & 3));

  // ---- buffer descriptors (4 SGPRs each): A relative to the first image of this tile, B = whole weight tensor.
  // The DMA is issued from inline asm: hipcc would otherwise wait vmcnt(0) before the first ds_read that follows
  // a pending LDS-DMA (it cannot prove the buffers distinct), which serialises the copy behind the MFMAs.
  const int img0 = MODE == 3 ? st_img : m0 / pix_per_img;
  const long long img_bytes = (long long)p.hi * p.wi * p.pix_stride * ESZ;
  const long long a_base_off = (long long)img0 * img_bytes;
  long long a_rem = p.in_bytes - a_base_off;
  if (a_rem > 0x7fffffffLL) a_rem = 0x7fffffffLL;
  const unsigned long long a_ptr = (unsigned long long)((const unsigned char*)p.in + a_base_off);
  const unsigned long long b_ptr = (unsigned long long)(sib ? p.wt2 : p.wt);
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)a_rem), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(sib ? p.wt2_bytes : p.wt_bytes), 0x00020000};
  // second operand pair of the fused stride-2 data gradient (same image geometry as `in`)
  const unsigned long long a2_ptr = (unsigned long long)((const unsigned char*)p.in2 + a_base_off);
  const unsigned long long b2_ptr = (unsigned long long)p.wt2;
  const i32x4 rs_a2 = {__builtin_amdgcn_readfirstlane((int)(unsigned)a2_ptr), __builtin_amdgcn_readfirstlane((int)((a2_ptr >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane((int)a_rem), 0x00020000};
  const i32x4 rs_b2 = {__builtin_amdgcn_readfirstlane((int)(unsigned)b2_ptr), __builtin_amdgcn_readfirstlane((int)((b2_ptr >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane(p.wt2_bytes), 0x00020000};

  // ---- per-row gather state (AROWS A rows per thread) ----
  uint32_t a_off[AROWS];
  int a_hb[AROWS], a_wb[AROWS];
  const int pixB = p.pix_stride * ESZ;
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + lrow + RPS * i;
    const bool ok = MODE == 3 || m < rows_total;
    const int mm = ok ? m : m0;
    int img = mm / pix_per_img;
    const int rem = mm - img * pix_per_img;
    int oh = rem / row_w;
    int ow = rem - oh * row_w;
    if constexpr (MODE == 2) {
      oh = 2 * oh + cls_ph;
      ow = 2 * ow + cls_pw;
    }
    if constexpr (MODE == 3) {  // patch pixel -> conv pixel, clamped into the image (clamped rows are never pooled)
      const int r_ = lrow + RPS * i;
      img = st_img;
      oh = min(max(14 * st_ty - 1 + (r_ >> 4), 0), p.ho - 1);
      ow = min(max(14 * st_tx - 1 + (r_ & 15), 0), p.wo - 1);
    }
    int hb, wb, hq, wq;
    if (FWD) {
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
    a_off[i] = (uint32_t)(((img - img0) * p.hi + hq) * p.wi + wq) * (uint32_t)pixB + (uint32_t)(lchunk_a * 16);
    a_hb[i] = ok ? hb : -(1 << 20);
    a_wb[i] = wb;
  }
  uint32_t b_off[BROWS];
  const int ktot_b = sib ? p.ci : p.ktot;  // K length of a weight row
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_off[i] = (uint32_t)(n0 + lrow + RPS * i) * (uint32_t)(ktot_b * ESZ) + (uint32_t)(lchunk_b * 16);

  uint32_t b_off2[BROWS];  // rows of the fused sibling's data-gradient weights [ci][1][1][ci2]
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_off2[i] = (uint32_t)(n0 + lrow + RPS * i) * (uint32_t)(p.ci2 * ESZ) + (uint32_t)(lchunk_b * 16);

  // LDS byte addresses (wave-uniform) of this wave's DMA pieces: piece i of an operand covers tile rows RPS i + 8 wave .. +7
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t lds_wave = lds_base + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * (8 * 128);

#define VDQN_DMA4(V0, V1, V2, V3, LDS, RSRC, SOFF)                                                                  \
  asm volatile(                                                                                                     \
      "s_nop 4\n\t"                                                                                                 \
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %5, %6 offen lds\n\t"                                \
      "s_add_u32 m0, %4, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds"                                 \
      ::"v"(V0), "v"(V1), "v"(V2), "v"(V3), "s"(LDS), "s"(RSRC), "s"(SOFF), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
      : "memory", "scc")
#define VDQN_ISSUE_X(BUF, KR, KS, C0, KSTEP, RSA, RSB, BOFF)                                                                       \
  {                                                                                                                 \
    const int delta_ = (FWD         ? (((KR)*p.wi + (KS)) * p.pix_stride + (C0))                                    \
                        : MODE == 1 ? ((C0) - ((KR)*p.wi + (KS)) * p.pix_stride)                                    \
                                    : ((C0) - (((KR) >> 1) * p.wi + ((KS) >> 1)) * p.pix_stride)) *                 \
                       ESZ;                                                                                         \
    uint32_t vo_[AROWS];                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < AROWS; ++i_) {                                                          \
      bool ok_;                                                                                                     \
      if (FWD) {                                                                                                    \
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
    const int zero_ = 0;                                                                                            \
    VDQN_DMA4(vo_[0], vo_[1], vo_[2], vo_[3], la_, RSA, zero_);                                                    \
    if constexpr (AROWS == 8) {                                                                                     \
      const uint32_t la2_ = la_ + 4 * PSTR;                                                                         \
      VDQN_DMA4(vo_[AROWS - 4], vo_[AROWS - 3], vo_[AROWS - 2], vo_[AROWS - 1], la2_, RSA, zero_);                 \
    }                                                                                                               \
    const uint32_t lb_ = lds_wave + (uint32_t)(NS * BM * 128) + (uint32_t)(BUF) * (BN * 128);                       \
    const int so_ = (KSTEP)*128;                                                                                    \
    if constexpr (BROWS == 4) {                                                                                     \
      VDQN_DMA4(BOFF[0], BOFF[BROWS > 1 ? 1 : 0], BOFF[BROWS > 2 ? 2 : 0], BOFF[BROWS > 2 ? 3 : 0], lb_, RSB, so_); \
    } else if constexpr (BROWS == 2) {                                                                              \
      asm volatile(                                                                                                 \
          "s_nop 4\n\t"                                                                                             \
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %4 offen lds\n\t"                            \
          "s_add_u32 m0, %2, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds"                             \
          ::"v"(BOFF[0]), "v"(BOFF[BROWS > 1 ? 1 : 0]), "s"(lb_), "s"(RSB), "s"(so_), "n"(PSTR)                  \
          : "memory", "scc");                                                                                       \
    } else {                                                                                                        \
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"             \
                   ::"v"(BOFF[0]), "s"(lb_), "s"(RSB), "s"(so_)                                                   \
                   : "memory");                                                                                     \
    }                                                                                                               \
  }
#define VDQN_ISSUE(BUF, KR, KS, C0, KSTEP) VDQN_ISSUE_X(BUF, KR, KS, C0, KSTEP, rs_a, rs_b, b_off)
  // the next K-step of this tile: one of its own taps, or (fused stride-2 data gradient, behind them) a channel chunk of the
  // sibling 1x1's output gradient at the centre tap
#define VDQN_ISSUE_NEXT(BUF)                                                                      \
  {                                                                                               \
    if (MODE == 2 && nk2 > 0 && issued >= nk) {                                                   \
      VDQN_ISSUE_X(BUF, kr0, ks0, (issued - nk) * KC, issued - nk, rs_a2, rs_b2, b_off2)          \
    } else {                                                                                      \
      VDQN_ISSUE(BUF, ikr, iks, ic0, VDQN_WSTEP())                                                \
      VDQN_ADVANCE()                                                                              \
    }                                                                                             \
    ++issued;                                                                                     \
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

  // acc[f][j]: pixels f*16 + i16 (lane & 15), channels g*CPL + j*4 + reg (g = lane >> 4)
  f32x4 acc[4][NF];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;

  // ---- main loop: K-steps enumerate (r, s, c0) with c0 fastest ----
  int ikr = kr0, iks = ks0, ic0 = 0;  // coordinates of the next K-step to issue
  int issued = 0;
  u32x4 fa[2][2][4], fb[2][2][NF];  // [register set][K half][fragment]
  const unsigned char* a_rd = sA + (wr * 64 + i16) * 128;
  const unsigned char* b_rd = sB + (wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128;
  const int coff0 = ((g ^ (i16 & 7)) << 4), coff1 = (((g + 4) ^ (i16 & 7)) << 4);
#define VDQN_LOAD_FRAGS(SET, BUF)                                                                                        \
  {                                                                                                                      \
    const unsigned char* a_ = a_rd + (BUF) * (BM * 128);                                                                 \
    const unsigned char* b_ = b_rd + (BUF) * (BN * 128);                                                                 \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                                   \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(a_ + f_ * 16 * 128 + coff0);                                      \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(a_ + f_ * 16 * 128 + coff1);                                      \
    }                                                                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                  \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + coff0);                                       \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + coff1);                                       \
    }                                                                                                                    \
  }
#define VDQN_MFMA_ALL(SET)                                                                                               \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)                     \
      _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                \
    if constexpr (sizeof(T) == 2) {                                                                                      \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[SET][h_][j_]),                 \
                                                            __builtin_bit_cast(bf16x8, fa[SET][h_][f_]), acc[f_][j_], 0, 0, 0); \
    } else {                                                                                                             \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].x), __uint_as_float(fa[SET][h_][f_].x), acc[f_][j_], 0, 0, 0); \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].y), __uint_as_float(fa[SET][h_][f_].y), acc[f_][j_], 0, 0, 0); \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].z), __uint_as_float(fa[SET][h_][f_].z), acc[f_][j_], 0, 0, 0); \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].w), __uint_as_float(fa[SET][h_][f_].w), acc[f_][j_], 0, 0, 0); \
    }                                                                                                                    \
  }
  // one K-step: tile K's fragments are in register set CUR (their reads were issued one step ago)
#define VDQN_STEP(K, CUR, NXT)                                                                                           \
  {                                                                                                                      \
    /* own DMA pieces of tile K+1 landed; own fragment reads of tile K complete (its buffer is about to be refilled) */  \
    if constexpr (NS == 2) {                                                                                             \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                        \
    } else { /* tiles K+2 .. issued-1 may stay in flight (AROWS + BROWS DMA instructions each, retired in order) */      \
      const int ahead_ = issued - (K) - 2;                                                                               \
      if (ahead_ >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (AROWS + BROWS)) : "memory");             \
      else if (ahead_ == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AROWS + BROWS) : "memory");              \
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                   \
    }                                                                                                                    \
    /* tell the compiler set CUR is complete HERE, so it does not wait for it (and for the younger reads) later */      \
    asm volatile("" : "+v"(fa[CUR][0][0]), "+v"(fa[CUR][0][1]), "+v"(fa[CUR][0][2]), "+v"(fa[CUR][0][3]),                \
                      "+v"(fa[CUR][1][0]), "+v"(fa[CUR][1][1]), "+v"(fa[CUR][1][2]), "+v"(fa[CUR][1][3]));               \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[CUR][0][j_]), "+v"(fb[CUR][1][j_]));     \
    __builtin_amdgcn_s_barrier();                                                                                        \
    if (issued < nk_all) VDQN_ISSUE_NEXT((K) & (NS - 1))                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    /* unconditional (the last step reads a stale buffer) so that the reads sit in ONE block with the MFMAs: every */    \
    /* fragment read then issues behind an MFMA instead of in front of all of them (tools/probes/mfma_peak.hip) */       \
    VDQN_LOAD_FRAGS(NXT, ((K) + 1) & (NS - 1))                                                                           \
    VDQN_MFMA_ALL(CUR)                                                                                                   \
    VDQN_INTERLEAVE(8 + 2 * NF)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
  if constexpr (MODE == 3 && ESZ == 2) {
    // Stem, bf16: K-step kr of conv row py reads the packed rows of conv row py + 1 at K-step kr - 1, so ONE staged window of
    // 19 x 16 rows (window row wy*16 + px = 128 bytes at packed pixel (y0 + wy, x0 + px)) serves all four K-steps: the
    // fragments of step kr are read 16 rows further down (the swizzle key px & 7 does not move).  The window (38 KiB) and
    // the whole weight matrix (4 x 8 KiB) are staged once per tile — 70 KiB instead of 160 KiB through the L1 — and the
    // K loop runs without barriers or DMA.  (A persistent one-workgroup-per-CU variant with the weights resident and the
    // window double-buffered was slower: experiments/stem_persistent.hip.)
    static_assert(MODE != 3 || ESZ != 2 || (AROWS == 8 && BROWS == 2), "stem staging assumes 256 threads");
    constexpr int WROWS = 320;  // 19 * 16 = 304 window rows, rounded up to the 32-row staging pass
    unsigned char* sBw = smem + WROWS * 128;
    const int y0 = 14 * st_ty - 1, x0 = 14 * st_tx - 1;
    uint32_t vw[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int R = lrow + RPS * i;
      const int sy = y0 + (R >> 4);
      const bool ok = (R < 304) && ((unsigned)sy < (unsigned)p.hi);
      // a packed column of -1 / 115 (only read for conv columns that are never pooled) wraps inside the image or falls out
      // of the descriptor's range (zeros): harmless either way
      vw[i] = ok ? (uint32_t)((sy * p.wi + x0 + (R & 15)) * pixB + lchunk_a * 16) : kOob;
    }
    const int zero_ = 0;
    VDQN_DMA4(vw[0], vw[1], vw[2], vw[3], lds_wave, rs_a, zero_);
    const uint32_t lw1_ = lds_wave + 4 * PSTR, lw2_ = lds_wave + 8 * PSTR;
    VDQN_DMA4(vw[4], vw[5], vw[6], vw[7], lw1_, rs_a, zero_);
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, 0 offen lds\n\t"
        "s_add_u32 m0, %2, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds"
        ::"v"(vw[8]), "v"(vw[9]), "s"(lw2_), "s"(rs_a), "n"(PSTR)
        : "memory", "scc");
#pragma unroll
    for (int ks_ = 0; ks_ < 4; ++ks_) {  // weight rows lrow, lrow + 32 of K-step ks_
      const uint32_t lbw_ = lds_wave + (uint32_t)(WROWS * 128 + ks_ * (BN * 128));
      const int so_ = ks_ * 128;
      asm volatile(
          "s_nop 4\n\t"
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %4 offen lds\n\t"
          "s_add_u32 m0, %2, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds"
          ::"v"(b_off[0]), "v"(b_off[BROWS > 1 ? 1 : 0]), "s"(lbw_), "s"(rs_b), "s"(so_), "n"(PSTR)
          : "memory", "scc");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* aw_ = smem + (wr * 64 + i16) * 128;
    const unsigned char* bw_ = sBw + ((i16 >> 2) * CPL + (i16 & 3)) * 128;
#pragma unroll
    for (int kr_ = 0; kr_ < 4; ++kr_) {
#pragma unroll
      for (int f_ = 0; f_ < 4; ++f_) {
        fa[0][0][f_] = *reinterpret_cast<const u32x4*>(aw_ + (kr_ * 16 + f_ * 16) * 128 + coff0);
        fa[0][1][f_] = *reinterpret_cast<const u32x4*>(aw_ + (kr_ * 16 + f_ * 16) * 128 + coff1);
      }
#pragma unroll
      for (int j_ = 0; j_ < NF; ++j_) {
        fb[0][0][j_] = *reinterpret_cast<const u32x4*>(bw_ + kr_ * (BN * 128) + j_ * 4 * 128 + coff0);
        fb[0][1][j_] = *reinterpret_cast<const u32x4*>(bw_ + kr_ * (BN * 128) + j_ * 4 * 128 + coff1);
      }
      VDQN_MFMA_ALL(0)
    }
  } else {
    // prologue: tiles 0 and 1 in flight, wait for tile 0 only
    VDQN_ISSUE_NEXT(0)
    if constexpr (NS == 4) {
      for (int b = 1; b < NS && issued < nk_all; ++b) VDQN_ISSUE_NEXT(b)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // once per tile
    } else if (issued < nk_all) {
      VDQN_ISSUE_NEXT(1)
      if constexpr (AROWS + BROWS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (AROWS + BROWS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (AROWS + BROWS == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if constexpr (AROWS + BROWS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // tile 0 visible
    VDQN_LOAD_FRAGS(0, 0)
    for (int k = 0; k < nk_all; k += 2) {
      VDQN_STEP(k, 0, 1)
      if (k + 1 < nk_all) VDQN_STEP(k + 1, 1, 0)
    }
  }
#undef VDQN_ISSUE_NEXT
#undef VDQN_ISSUE_X
#undef VDQN_LOAD_FRAGS
#undef VDQN_MFMA_ALL
#undef VDQN_STEP
#undef VDQN_ISSUE
#undef VDQN_DMA4
#undef VDQN_ADVANCE
#undef VDQN_WSTEP

  if constexpr (MODE == 3) {
    // ---- stem epilogue: bias + ReLU, the 16x16x64 patch goes to LDS (16-byte chunks XOR-swizzled by the pixel), then
    // 49 pooled pixels x 64 channels are reduced from it with the first-maximum-wins rule of maxpool_fwd_kernel ----
    static_assert(MODE != 3 || (BM == 256 && BN == 64 && WN == 1), "stem tile is 16x16 pixels x 64 channels");
    constexpr int E16 = 16 / ESZ, CPP = 64 / E16;  // elements per 16 bytes; chunks per pixel
    __syncthreads();  // every wave is past its last fragment read
    T* sT = reinterpret_cast<T*>(smem);
    {
      float bv[CPL];
#pragma unroll
      for (int e = 0; e < CPL; ++e) bv[e] = p.bias[g * CPL + e];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int r = wr * 64 + f * 16 + i16;
        T ov[CPL];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) ov[j * 4 + q] = from_f32<T>(fmaxf(acc[f][j][q] + bv[j * 4 + q], 0.f));
#pragma unroll
        for (int c = 0; c < CPL / E16; ++c) {
          const int chunk = (g * (CPL / E16) + c) ^ (r & 7);
          *reinterpret_cast<uint4*>(sT + (size_t)r * 64 + chunk * E16) = reinterpret_cast<const uint4*>(ov)[c];
        }
      }
    }
    __syncthreads();
    T* __restrict__ pool = (T*)p.pool_out;
    for (int item = tid; item < 49 * CPP; item += NT) {
      const int pp = item / CPP, cg = item - pp * CPP;
      const int pi = pp / 7, pj = pp - pi * 7;
      float best[E16];
      uint8_t bi[E16];
      if constexpr (ESZ == 2) {
        // post-ReLU bf16 values are >= +0 (or NaN), so their bit patterns order like unsigned integers with NaN on top;
        // key = bits << 4 | (8 - tap) makes ONE v_max_u32 per element and tap do "larger value, else earlier tap"
        uint32_t key[E16];
#pragma unroll
        for (int e = 0; e < E16; ++e) key[e] = 0u;
        bool any = false;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int py = 2 * pi + kh, y = 14 * st_ty - 1 + py;
          if ((unsigned)y >= (unsigned)p.ho) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int px = 2 * pj + kw, x = 14 * st_tx - 1 + px;
            if ((unsigned)x >= (unsigned)p.wo) continue;
            const int r = py * 16 + px;
            const uint4 v = *reinterpret_cast<const uint4*>(sT + (size_t)r * 64 + ((cg ^ (r & 7)) * E16));
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            const uint32_t tag = (uint32_t)(8 - (kh * 3 + kw));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              key[2 * q] = max(key[2 * q], ((w4[q] << 4) & 0x7fff0u) | tag);  // sign bit dropped: ReLU may leave -0.0
              key[2 * q + 1] = max(key[2 * q + 1], ((w4[q] >> 12) & 0x7fff0u) | tag);
            }
            any = true;
          }
        }
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          best[e] = any ? bf16_to_f32((bf16raw)(key[e] >> 4)) : -INFINITY;
          bi[e] = any ? (uint8_t)(8u - (key[e] & 15u)) : (uint8_t)0;
        }
      } else {
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          best[e] = -INFINITY;
          bi[e] = 0;
        }
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int py = 2 * pi + kh, y = 14 * st_ty - 1 + py;
          if ((unsigned)y >= (unsigned)p.ho) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int px = 2 * pj + kw, x = 14 * st_tx - 1 + px;
            if ((unsigned)x >= (unsigned)p.wo) continue;
            const int r = py * 16 + px;
            const uint4 v = *reinterpret_cast<const uint4*>(sT + (size_t)r * 64 + ((cg ^ (r & 7)) * E16));
            const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int e = 0; e < E16; ++e) {
              const float fv = to_f32<T>(pv[e]);
              if (first || fv > best[e] || fv != fv) {
                best[e] = fv;
                bi[e] = (uint8_t)(kh * 3 + kw);
              }
            }
            first = false;
          }
        }
      }
      const size_t o = (((size_t)st_img * 56 + 7 * st_ty + pi) * 56 + 7 * st_tx + pj) * 64 + cg * E16;
      T ov[E16];
#pragma unroll
      for (int e = 0; e < E16; ++e) ov[e] = from_f32<T>(best[e]);
      *reinterpret_cast<uint4*>(pool + o) = *reinterpret_cast<const uint4*>(ov);
      if (st_img < p.n_idx_img) {  // the other images: pooled values only (vdqn_stem_conv_pool_n)
        if constexpr (E16 == 8) *reinterpret_cast<uint2*>(p.pool_idx + o) = *reinterpret_cast<const uint2*>(bi);
        else *reinterpret_cast<uint32_t*>(p.pool_idx + o) = *reinterpret_cast<const uint32_t*>(bi);
      }
    }
    return;
  }
  if (sib) {  // the sibling's own epilogue: its bias, ReLU flag and output tensor; no residual, mask or column sums
    IgemmParams q = p;
    q.bias = p.bias2; q.out = p.out2; q.relu = p.relu2; q.co = p.co2; q.ldo = p.ldo2;
    q.resid = nullptr; q.mask = nullptr; q.out_f32 = nullptr; q.colsum_part = nullptr;
    igemm_epilogue<T, BM, BN, MODE, WN>(q, acc, smem, m0, n0, tile_m, rows_total, pix_per_img, row_w, cls_ph, cls_pw, q.bias);
    return;
  }
  if constexpr (MODE == 2 && sizeof(T) == 2 && BM == 128 && WN == 2) {
    // stride-2 data gradient, bf16: the lean epilogue (igemm_common.h) — these tiles have 2-8 K-steps, the shared epilogue was as
    // long as their K loop.  Class-local row -> output pixel as in igemm_epilogue.
    const bool lean_d = !p.no_lean && p.vec_ok && p.out && !p.out_f32 && !p.bias && p.co % BN == 0 && (long long)p.M * p.ldo * 2 < 0x7fffffffLL;
    if (lean_d) {
      const LeanEpiD led = make_lean_epi_d(p.out, p.resid, p.mask, p.colsum_part, p.M, p.ldo, p.co);
      const int ncol = n0 + wc * (BN / WN) + g * CPL;
      uint32_t off[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int m = m0 + wr * 64 + f * 16 + i16;
        const bool ok = m < rows_total && ncol < p.co;
        const int mm = ok ? m : m0;
        const int img = mm / pix_per_img;
        const int rem = mm - img * pix_per_img;
        const int ohc = rem / row_w;
        const int mo = (img * p.ho + 2 * ohc + cls_ph) * p.wo + 2 * (rem - ohc * row_w) + cls_pw;
        off[f] = ok ? (uint32_t)(mo * p.ldo + ncol) * 2u : kOob;
      }
      lean_epilogue_dgrad<NF>(led, acc, reinterpret_cast<float*>(smem), off, n0, tile_m, tid);
      return;
    }
  }
  igemm_epilogue<T, BM, BN, MODE, WN>(p, acc, smem, m0, n0, tile_m, rows_total, pix_per_img, row_w, cls_ph, cls_pw, p.bias);
}

// ---------------------------------------------------------------------------------------------------------
// Window variant for 3x3 / stride 1 / pad 1 convolutions (MODE 0 forward, MODE 1 data gradient), 128 x BN tiles.
// The three horizontal taps of one kernel row read the same input pixels shifted by one, so the activation operand of a
// (kernel row, channel chunk) is staged ONCE as a window of BM + 8 consecutive pixels (window row j = output pixel
// m0 - 1 + j at the centre tap) and the fragments of tap ks are read at a row offset of 0 / 1 / 2; lanes whose neighbour
// would wrap around an image row (ow = 0 for the left tap, ow = W - 1 for the right one) read a zero row instead.  The
// activation bytes that go L2 -> LDS drop by 3x (the weights still stream per tap): 37 % less staging for 128x128 tiles,
// 55 % less for the 64-channel layers.  K-steps run (kr, c0, ks) with ks innermost; everything else — LDS-DMA from inline
// asm, source-side swizzle (key = window row & 7), register double-buffered fragments, one barrier per K-step, the
// LDS-free epilogue — is the generic kernel's.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int BN, int MODE>
__global__ __launch_bounds__(256, BN == 64 ? 3 : 2) void igemm_win_kernel(const IgemmParams p) {
  static_assert(MODE == 0 || MODE == 1, "window kernel: forward or stride-1 data gradient");
  constexpr int BM = 128, WN = 2, RPS = 32, AROWS = 4;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int KC = 128 / ESZ;
  constexpr int NF = BN / (16 * WN);
  constexpr int CPL = 4 * NF;
  constexpr int BROWS = BN / RPS;
  constexpr int PSTR = RPS * 128;
  constexpr int WROWS = BM + 8;            // window rows (BM + 2 needed; the 8-row DMA piece is the granule)
  constexpr int WBYTES = WROWS * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                          // [2][WROWS][128 B]
  unsigned char* sB = smem + 2 * WBYTES;             // [2][BN][128 B]
  unsigned char* sZ = smem + 2 * WBYTES + 2 * BN * 128;  // one row of zeros

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lb % (uint32_t)p.tiles_n), tile_m = (int)(lb / (uint32_t)p.tiles_n);
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int nk = p.nk;
  const int row_w = p.wo, pix_per_img = p.howo, rows_total = p.M;
  const int lrow = tid >> 3;
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int lchunk_b = (tid & 7) ^ ((((lrow / CPL) & 1) << 2) | (lrow & 3));
  if (tid < 8) reinterpret_cast<uint4*>(sZ)[tid] = make_uint4(0, 0, 0, 0);

  const int q0 = m0 > 0 ? m0 - 1 : 0;  // first pixel of the window that exists
  const int img0 = q0 / pix_per_img;
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

  // ---- window rows staged by this thread: j = lrow + 32 i (i < 4), and j = BM + lrow for the first wave ----
  const int pixB = p.pix_stride * ESZ;
  uint32_t a_off[AROWS + 1];
  int a_hb[AROWS + 1];
#pragma unroll
  for (int i = 0; i <= AROWS; ++i) {
    const int j = i < AROWS ? lrow + RPS * i : BM + (lrow & 7);
    const int q = m0 - 1 + j;
    const bool ok = q >= 0 && q < rows_total;
    const int qq = ok ? q : q0;
    const int img = qq / pix_per_img;
    const int rem = qq - img * pix_per_img;
    const int oh = rem / row_w, ow = rem - oh * row_w;
    // centre-tap source pixel: forward (oh - 1 + kr, ow), data gradient (oh + 1 - kr, ow)
    const int hb = MODE == 0 ? oh - 1 : oh + 1;
    a_off[i] = (uint32_t)(((img - img0) * p.hi + hb) * p.wi + ow) * (uint32_t)pixB + (uint32_t)(lchunk_a * 16);
    a_hb[i] = ok ? hb : -(1 << 20);
  }
  uint32_t b_off[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_off[i] = (uint32_t)(n0 + lrow + RPS * i) * (uint32_t)(p.ktot * ESZ) + (uint32_t)(lchunk_b * 16);

  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * (8 * 128);

#define VDQN_DMA4(V0, V1, V2, V3, LDS, RSRC, SOFF)                                                                  \
  asm volatile(                                                                                                     \
      "s_nop 4\n\t"                                                                                                 \
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %5, %6 offen lds\n\t"                                \
      "s_add_u32 m0, %4, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds"                                 \
      ::"v"(V0), "v"(V1), "v"(V2), "v"(V3), "s"(LDS), "s"(RSRC), "s"(SOFF), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
      : "memory", "scc")
  // activation window of (kernel row KR, channel chunk C0) -> window buffer WBUF
#define VDQN_ISSUE_AW(WBUF, KR, C0)                                                                                 \
  {                                                                                                                 \
    const int delta_ = (MODE == 0 ? ((KR)*p.wi * p.pix_stride + (C0)) : ((C0) - (KR)*p.wi * p.pix_stride)) * ESZ;   \
    uint32_t vo_[AROWS + 1];                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ <= AROWS; ++i_) {                                                         \
      const bool ok_ = MODE == 0 ? ((unsigned)(a_hb[i_] + (KR)) < (unsigned)p.hi) : ((unsigned)(a_hb[i_] - (KR)) < (unsigned)p.hi); \
      vo_[i_] = ok_ ? a_off[i_] + (uint32_t)delta_ : kOob;                                                          \
    }                                                                                                               \
    const uint32_t la_ = lds_wave + (uint32_t)(WBUF) * WBYTES;                                                      \
    const int zero_ = 0;                                                                                            \
    VDQN_DMA4(vo_[0], vo_[1], vo_[2], vo_[3], la_, rs_a, zero_);                                                    \
    if (wave_u == 0) {                                                                                              \
      const uint32_t lx_ = lds_base + (uint32_t)(WBUF) * WBYTES + BM * 128;                                         \
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"              \
                   ::"v"(vo_[AROWS]), "s"(lx_), "s"(rs_a)                                                           \
                   : "memory");                                                                                     \
    }                                                                                                               \
  }
  // weight tile of K-step KSTEP -> weight buffer BUF
#define VDQN_ISSUE_B(BUF, KSTEP)                                                                                    \
  {                                                                                                                 \
    const uint32_t lb_ = lds_wave + (uint32_t)(2 * WBYTES) + (uint32_t)(BUF) * (BN * 128);                          \
    const int so_ = (KSTEP)*128;                                                                                    \
    if constexpr (BROWS == 4) {                                                                                     \
      VDQN_DMA4(b_off[0], b_off[BROWS > 1 ? 1 : 0], b_off[BROWS > 2 ? 2 : 0], b_off[BROWS > 2 ? 3 : 0], lb_, rs_b, so_); \
    } else {                                                                                                        \
      asm volatile(                                                                                                 \
          "s_nop 4\n\t"                                                                                             \
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %4 offen lds\n\t"                            \
          "s_add_u32 m0, %2, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds"                             \
          ::"v"(b_off[0]), "v"(b_off[BROWS > 1 ? 1 : 0]), "s"(lb_), "s"(rs_b), "s"(so_), "n"(PSTR)                  \
          : "memory", "scc");                                                                                       \
    }                                                                                                               \
  }

  f32x4 acc[4][NF];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  // edge lanes: bit f of e_left / e_right = pixel (wr*64 + f*16 + i16) sits in image column 0 / W - 1
  uint32_t e_left = 0, e_right = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int m = m0 + wr * 64 + f * 16 + i16;
    const int ow = (m % pix_per_img) % row_w;
    e_left |= (ow == 0 ? 1u : 0u) << f;
    e_right |= (ow == row_w - 1 ? 1u : 0u) << f;
  }

  const int cpk = p.ci / KC;           // channel chunks per tap
  // coordinates of the next K-step to ISSUE and of the next K-step whose fragments are LOADED: (kr, chunk, ks), ks fastest
  int i_kr = 0, i_cc = 0, i_ks = 0, i_grp = 0, issued = 0;
  int l_ks = 0, l_grp = 0;
#define VDQN_ADV(KR, CC, KS, GRP) \
  {                               \
    if (++(KS) == 3) {            \
      (KS) = 0;                   \
      ++(GRP);                    \
      if (++(CC) == cpk) {        \
        (CC) = 0;                 \
        ++(KR);                   \
      }                           \
    }                             \
  }
#define VDQN_ISSUE_STEP(BBUF)                                              \
  {                                                                        \
    VDQN_ISSUE_B(BBUF, (i_kr * 3 + i_ks) * cpk + i_cc)                     \
    if (i_ks == 0) VDQN_ISSUE_AW(i_grp & 1, i_kr, i_cc * KC)               \
    VDQN_ADV(i_kr, i_cc, i_ks, i_grp)                                      \
    ++issued;                                                              \
  }

  u32x4 fa[2][2][4], fb[2][2][NF];
  const unsigned char* a_rd = sA + (wr * 64 + i16) * 128;
  const unsigned char* b_rd = sB + (wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128;
  const int bcoff0 = ((g ^ (i16 & 7)) << 4), bcoff1 = (((g + 4) ^ (i16 & 7)) << 4);
  const unsigned char* z_rd = sZ + (g << 4);
  // fragments of the K-step (l_grp, l_ks): window row = tile row + dxi, dxi = ks (forward) / 2 - ks (data gradient)
#define VDQN_LOAD_FRAGS(SET, BBUF)                                                                                       \
  {                                                                                                                      \
    const int dxi_ = MODE == 0 ? l_ks : 2 - l_ks;                                                                        \
    const uint32_t zm_ = dxi_ == 0 ? e_left : (dxi_ == 2 ? e_right : 0u);                                                \
    const int key_ = (i16 + dxi_) & 7;                                                                                   \
    const int ac0_ = ((g ^ key_) << 4), ac1_ = (((g + 4) ^ key_) << 4);                                                  \
    const unsigned char* a_ = a_rd + (l_grp & 1) * WBYTES + dxi_ * 128;                                                  \
    const unsigned char* b_ = b_rd + (BBUF) * (BN * 128);                                                                \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                                   \
      const bool z_ = (zm_ >> f_) & 1u;                                                                                  \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(z_ ? z_rd : a_ + f_ * 16 * 128 + ac0_);                          \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(z_ ? z_rd + 64 : a_ + f_ * 16 * 128 + ac1_);                     \
    }                                                                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                  \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + bcoff0);                                      \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + bcoff1);                                      \
    }                                                                                                                    \
    if (++l_ks == 3) {                                                                                                   \
      l_ks = 0;                                                                                                          \
      ++l_grp;                                                                                                           \
    }                                                                                                                    \
  }
#define VDQN_MFMA_ALL(SET)                                                                                               \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)                     \
      _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                \
    if constexpr (sizeof(T) == 2) {                                                                                      \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[SET][h_][j_]),                 \
                                                            __builtin_bit_cast(bf16x8, fa[SET][h_][f_]), acc[f_][j_], 0, 0, 0); \
    } else {                                                                                                             \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].x), __uint_as_float(fa[SET][h_][f_].x), acc[f_][j_], 0, 0, 0); \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].y), __uint_as_float(fa[SET][h_][f_].y), acc[f_][j_], 0, 0, 0); \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].z), __uint_as_float(fa[SET][h_][f_].z), acc[f_][j_], 0, 0, 0); \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fb[SET][h_][j_].w), __uint_as_float(fa[SET][h_][f_].w), acc[f_][j_], 0, 0, 0); \
    }                                                                                                                    \
  }
#define VDQN_STEP(K, CUR, NXT)                                                                                           \
  {                                                                                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                          \
    asm volatile("" : "+v"(fa[CUR][0][0]), "+v"(fa[CUR][0][1]), "+v"(fa[CUR][0][2]), "+v"(fa[CUR][0][3]),                \
                      "+v"(fa[CUR][1][0]), "+v"(fa[CUR][1][1]), "+v"(fa[CUR][1][2]), "+v"(fa[CUR][1][3]));               \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[CUR][0][j_]), "+v"(fb[CUR][1][j_]));     \
    __builtin_amdgcn_s_barrier();                                                                                        \
    if (issued < nk) VDQN_ISSUE_STEP((K) & 1)                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    VDQN_LOAD_FRAGS(NXT, ((K) + 1) & 1) /* unconditional (the last step reads a stale buffer): one block with the MFMAs */ \
    VDQN_MFMA_ALL(CUR)                                                                                                   \
    VDQN_INTERLEAVE(8 + 2 * NF)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
  // prologue: K-steps 0 and 1 (window of group 0 + two weight tiles)
  VDQN_ISSUE_STEP(0)
  if (issued < nk) VDQN_ISSUE_STEP(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // window 0, weight tiles 0/1 and the zero row are visible
  VDQN_LOAD_FRAGS(0, 0)
  for (int k = 0; k < nk; k += 2) {
    VDQN_STEP(k, 0, 1)
    if (k + 1 < nk) VDQN_STEP(k + 1, 1, 0)
  }
#undef VDQN_LOAD_FRAGS
#undef VDQN_MFMA_ALL
#undef VDQN_STEP
#undef VDQN_ISSUE_STEP
#undef VDQN_ISSUE_AW
#undef VDQN_ISSUE_B
#undef VDQN_ADV
#undef VDQN_DMA4
  igemm_epilogue<T, BM, BN, MODE, WN>(p, acc, smem, m0, n0, tile_m, rows_total, pix_per_img, row_w, 0, 0, p.bias);
}

template <typename T, int BN, int MODE>
int launch_igemm_win(const IgemmParams& p, hipStream_t stream) {
  const size_t smem = 2 * (128 + 8) * 128 + 2 * BN * 128 + 128;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&igemm_win_kernel<T, BN, MODE>), (size_t)smem);
  const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
  const double esz = sizeof(T);
  static const char* const kTag[2][2][2] = {{{"igemm_win<bf16,64,fwd>", "igemm_win<bf16,64,dgrad>"}, {"igemm_win<bf16,128,fwd>", "igemm_win<bf16,128,dgrad>"}},
                                            {{"igemm_win<f32,64,fwd>", "igemm_win<f32,64,dgrad>"}, {"igemm_win<f32,128,fwd>", "igemm_win<f32,128,dgrad>"}}};
  vdqn_prof_begin(kTag[sizeof(T) == 2 ? 0 : 1][BN == 128 ? 1 : 0][MODE], 2.0 * p.M * p.co * p.ktot,
                  esz * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))),
                  stream);
  hipLaunchKernelGGL((igemm_win_kernel<T, BN, MODE>), dim3(grid), dim3(256), smem, stream, p);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Nine-tap window variant (bf16, 128 x 128 tiles, images up to 28 pixels wide: layer2 - layer4).  igemm_win_kernel stages one
// window per (kernel row, channel chunk) = three per nine K-steps; here ONE window of 128 + 2 W + 2 consecutive input pixels
// (window row j = pixel m0 - W - 1 + j, all images laid end to end) per channel chunk serves all nine taps: tap (ky, kx) of tile
// row r reads window row r + W ky + kx, and lanes whose tap leaves the image (first / last row or column) read a zero row — the
// scheme of conv64_kernel with the weights still streamed per K-step.  K order (chunk, kr, ks).  Staged activation bytes per
// nine K-steps: 20-24 KiB instead of 51 KiB (the weight tiles, 144 KiB, are unchanged): 12-16 % less L2 -> LDS traffic, which
// is what bounds these kernels (experiments/README.md).  Everything else is igemm_win_kernel's.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void igemm_win9_kernel(const IgemmParams p, const int wrows, const FastDiv d_wo, const FastDiv d_howo) {
  static_assert(MODE == 0 || MODE == 1, "window kernel: forward or stride-1 data gradient");
  constexpr int BM = 128, BN = 128, WN = 2, RPS = 32;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int KC = 128 / ESZ;
  constexpr int NF = BN / (16 * WN);
  constexpr int CPL = 4 * NF;
  constexpr int BROWS = BN / RPS;
  constexpr int PSTR = RPS * 128;
  constexpr int NPASS = 6;                 // 32-row staging passes that cover the largest window (192 rows)
  const int WBYTES = wrows * 128;          // wrows: multiple of 8, > 128 + 2 W + 2 (its last row is never sourced: the zero row)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                          // [2][wrows][128 B]
  unsigned char* sB = smem + 2 * WBYTES;             // [2][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
#ifdef VDQN_PRIO
  {  // experiment: the wave in the odd slot of its SIMD runs at priority 1 (anti-phase of the two co-resident workgroups)
    unsigned hwid_p;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_p));
    if (hwid_p & 1) __builtin_amdgcn_s_setprio(VDQN_PRIO);
  }
#endif
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lb % (uint32_t)p.tiles_n), tile_m = (int)(lb / (uint32_t)p.tiles_n);
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int nk = p.nk;
  const int W = p.wo, H = p.ho, rows_total = p.M;
  const int lrow = tid >> 3;
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int lchunk_b = (tid & 7) ^ ((((lrow / CPL) & 1) << 2) | (lrow & 3));

  const unsigned long long a_ptr = (unsigned long long)p.in;
  const unsigned long long b_ptr = (unsigned long long)p.wt;
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};

  // ---- window rows staged by this thread: j = lrow + 32 i; rows past 128 + 2 W + 2 (and pixels outside the tensor) are zero ----
  const int pixB = p.pix_stride * ESZ;
  const int need = BM + 2 * W + 2;
  uint32_t a_off[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int j = lrow + RPS * i;
    const int q = m0 - W - 1 + j;
    a_off[i] = (j < need && (unsigned)q < (unsigned)rows_total) ? (uint32_t)q * (uint32_t)pixB + (uint32_t)(lchunk_a * 16) : kOob;
  }
  uint32_t b_off[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_off[i] = (uint32_t)(n0 + lrow + RPS * i) * (uint32_t)(p.ktot * ESZ) + (uint32_t)(lchunk_b * 16);

  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * (8 * 128);
  const int npass_w = (wrows - 8 * wave_u + 31) / 32;  // staging passes in which this wave's 8-row piece exists (wave-uniform)

#define VDQN_DMA1(V0, LDS, RSRC, SOFF)                                                                             \
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" ::"v"(V0), "s"(LDS), "s"(RSRC), "s"(SOFF) : "memory")
#define VDQN_DMA4(V0, V1, V2, V3, LDS, RSRC, SOFF)                                                                  \
  asm volatile(                                                                                                     \
      "s_nop 4\n\t"                                                                                                 \
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %5, %6 offen lds\n\t"                                \
      "s_add_u32 m0, %4, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds"                                 \
      ::"v"(V0), "v"(V1), "v"(V2), "v"(V3), "s"(LDS), "s"(RSRC), "s"(SOFF), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
      : "memory", "scc")
  // activation window of channel chunk CC -> window buffer WBUF: passes 0..3 always exist (wrows >= 144), 4 and 5 per wave
#define VDQN_ISSUE_AW(WBUF, CC)                                                                                     \
  {                                                                                                                 \
    const uint32_t la_ = lds_wave + (uint32_t)(WBUF) * (uint32_t)WBYTES;                                            \
    const int so_a_ = (CC) * 128;                                                                                   \
    VDQN_DMA4(a_off[0], a_off[1], a_off[2], a_off[3], la_, rs_a, so_a_);                                            \
    if (npass_w > 4) {                                                                                              \
      const uint32_t l4_ = la_ + 4 * PSTR;                                                                          \
      VDQN_DMA1(a_off[4], l4_, rs_a, so_a_);                                                                        \
    }                                                                                                               \
    if (npass_w > 5) {                                                                                              \
      const uint32_t l5_ = la_ + 5 * PSTR;                                                                          \
      VDQN_DMA1(a_off[5], l5_, rs_a, so_a_);                                                                        \
    }                                                                                                               \
  }
#define VDQN_ISSUE_B(BUF, KSTEP)                                                                                    \
  {                                                                                                                 \
    const uint32_t lb_ = lds_wave + (uint32_t)(2 * WBYTES) + (uint32_t)(BUF) * (BN * 128);                          \
    const int so_ = (KSTEP)*128;                                                                                    \
    VDQN_DMA4(b_off[0], b_off[1], b_off[2], b_off[3], lb_, rs_b, so_);                                              \
  }

  f32x4 acc[4][NF];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  // edge bits of this lane's four pixels, 4 bits per fragment f: 1 top row, 2 bottom row, 4 left column, 8 right column
  uint32_t edge16 = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const uint32_t m = (uint32_t)(m0 + wr * 64 + f * 16 + i16);
    const uint32_t rem = m - fastdiv(m, d_howo) * d_howo.div;
    const uint32_t oh = fastdiv(rem, d_wo), ow = rem - oh * d_wo.div;
    const uint32_t e = (oh == 0 ? 1u : 0u) | (oh == (uint32_t)H - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == (uint32_t)W - 1 ? 8u : 0u);
    edge16 |= e << (4 * f);
  }

  const int cpk = p.ci / KC;           // channel chunks
  // next K-step to ISSUE and next K-step whose fragments are LOADED: (chunk, tap), tap fastest
  int i_cc = 0, i_tap = 0, issued = 0;
  int l_cc = 0, l_tap = 0;
#define VDQN_ISSUE_STEP(BBUF)                                              \
  {                                                                        \
    VDQN_ISSUE_B(BBUF, i_tap * cpk + i_cc)                                 \
    if (i_tap == 0) VDQN_ISSUE_AW(i_cc & 1, i_cc)                          \
    if (++i_tap == 9) {                                                    \
      i_tap = 0;                                                           \
      ++i_cc;                                                              \
    }                                                                      \
    ++issued;                                                              \
  }

  u32x4 fa[2][2][4], fb[2][2][NF];
  const unsigned char* a_rd = sA + (wr * 64 + i16) * 128;
  const unsigned char* b_rd = sB + (wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128;
  const int bcoff0 = ((g ^ (i16 & 7)) << 4), bcoff1 = (((g + 4) ^ (i16 & 7)) << 4);
  const int zoff = (wrows - 1) * 128 + (g << 4);  // the zero row of a window buffer
  // fragments of the K-step (l_cc, l_tap): the forward reads input pixel m + (kr-1) W + (ks-1), the data gradient m + (1-kr) W + (1-ks)
#define VDQN_LOAD_FRAGS(SET, BBUF)                                                                                       \
  {                                                                                                                      \
    const int kr_ = (l_tap * 11) >> 5, ks_ = l_tap - 3 * kr_; /* l_tap / 3 for 0..8 */                                   \
    const int ky_ = MODE == 0 ? kr_ : 2 - kr_, kx_ = MODE == 0 ? ks_ : 2 - ks_;                                          \
    const uint32_t tb_ = (ky_ == 0 ? 1u : 0u) | (ky_ == 2 ? 2u : 0u) | (kx_ == 0 ? 4u : 0u) | (kx_ == 2 ? 8u : 0u);       \
    const uint32_t zm_ = edge16 & (tb_ * 0x1111u);                                                                       \
    const int off_ = W * ky_ + kx_;                                                                                      \
    const int key_ = (i16 + off_) & 7;                                                                                   \
    const int ac0_ = ((g ^ key_) << 4), ac1_ = (((g + 4) ^ key_) << 4);                                                  \
    const unsigned char* w_ = sA + (l_cc & 1) * WBYTES;                                                                  \
    const unsigned char* a_ = a_rd + (l_cc & 1) * WBYTES + off_ * 128;                                                   \
    const unsigned char* b_ = b_rd + (BBUF) * (BN * 128);                                                                \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                                   \
      const bool z_ = ((zm_ >> (4 * f_)) & 15u) != 0u;                                                                   \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(z_ ? w_ + zoff : a_ + f_ * 16 * 128 + ac0_);                      \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(z_ ? w_ + zoff + 64 : a_ + f_ * 16 * 128 + ac1_);                 \
    }                                                                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                  \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + bcoff0);                                      \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + bcoff1);                                      \
    }                                                                                                                    \
    if (++l_tap == 9) {                                                                                                  \
      l_tap = 0;                                                                                                         \
      ++l_cc;                                                                                                            \
    }                                                                                                                    \
  }
#define VDQN_MFMA_ALL(SET)                                                                                               \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)                     \
      _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                \
    acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[SET][h_][j_]),                   \
                                                          __builtin_bit_cast(bf16x8, fa[SET][h_][f_]), acc[f_][j_], 0, 0, 0); \
  }
#ifdef VDQN_STAMP
  // diagnostic build only (tools/stamp_win9.py): s_memtime around the phases of every K-step, summed per workgroup by wave 0
  unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_comp = 0, st_t = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long st_rt_begin = __builtin_amdgcn_s_memrealtime();
#define VDQN_ST(ACC)                                                     \
  {                                                                      \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();          \
    ACC += n_ - st_t;                                                    \
    st_t = n_;                                                           \
  }
#else
#define VDQN_ST(ACC)
#endif
#define VDQN_STEP(K, CUR, NXT)                                                                                           \
  {                                                                                                                      \
    VDQN_ST(st_comp)                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                          \
    VDQN_ST(st_wait)                                                                                                     \
    asm volatile("" : "+v"(fa[CUR][0][0]), "+v"(fa[CUR][0][1]), "+v"(fa[CUR][0][2]), "+v"(fa[CUR][0][3]),                \
                      "+v"(fa[CUR][1][0]), "+v"(fa[CUR][1][1]), "+v"(fa[CUR][1][2]), "+v"(fa[CUR][1][3]));               \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[CUR][0][j_]), "+v"(fb[CUR][1][j_]));     \
    __builtin_amdgcn_s_barrier();                                                                                        \
    VDQN_ST(st_bar)                                                                                                      \
    if (issued < nk) VDQN_ISSUE_STEP((K) & 1)                                                                            \
    VDQN_ST(st_issue)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    VDQN_LOAD_FRAGS(NXT, ((K) + 1) & 1) /* unconditional: the step behind the last one re-reads buffers that still exist */ \
    VDQN_MFMA_ALL(CUR)                                                                                                   \
    VDQN_INTERLEAVE(8 + 2 * NF)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
  VDQN_ISSUE_STEP(0)
  if (issued < nk) VDQN_ISSUE_STEP(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // window 0 and weight tiles 0 / 1 are visible
  VDQN_LOAD_FRAGS(0, 0)
#ifdef VDQN_STAMP
  st_t = __builtin_amdgcn_s_memtime();
#endif
  for (int k = 0; k < nk; k += 2) {
    VDQN_STEP(k, 0, 1)
    if (k + 1 < nk) VDQN_STEP(k + 1, 1, 0)
  }
  VDQN_ST(st_comp)
#undef VDQN_ST
#undef VDQN_LOAD_FRAGS
#undef VDQN_MFMA_ALL
#undef VDQN_STEP
#undef VDQN_ISSUE_STEP
#undef VDQN_ISSUE_AW
#undef VDQN_ISSUE_B
#undef VDQN_DMA4
#undef VDQN_DMA1
#ifdef VDQN_STAMP
  const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
  igemm_epilogue<T, BM, BN, MODE, WN>(p, acc, smem, m0, n0, tile_m, rows_total, p.howo, W, 0, 0, p.bias);
#ifdef VDQN_STAMP
  if (p.pool_out && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the epilogue's stores have left
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.pool_out) + (size_t)blockIdx.x * 16;
    o[0] = st_begin; o[1] = st_loop_end; o[2] = __builtin_amdgcn_s_memtime();
    o[3] = st_wait; o[4] = st_bar; o[5] = st_issue; o[6] = st_comp; o[7] = (unsigned long long)nk;
    o[8] = st_rt_begin; o[9] = __builtin_amdgcn_s_memrealtime();  // 100 MHz reference clock
    unsigned hwid_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_));
    o[10] = hwid_;
  }
#endif
}

#ifdef VDQN_STAMP
}  // namespace
void* g_stamp_buffer = nullptr;
namespace {
#endif

template <typename T, int MODE>
int launch_igemm_win9(const IgemmParams& p, hipStream_t stream) {
  const int wrows = (128 + 2 * p.wo + 2 + 1 + 7) & ~7;
  const size_t smem = (size_t)2 * wrows * 128 + 2 * 128 * 128;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&igemm_win9_kernel<T, MODE>), smem);
  const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
  vdqn_prof_begin(MODE == 0 ? "igemm_win<bf16,128,fwd>" : "igemm_win<bf16,128,dgrad>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))), stream);
#ifdef VDQN_STAMP
  IgemmParams ps = p;
  ps.pool_out = g_stamp_buffer;
  hipLaunchKernelGGL((igemm_win9_kernel<T, MODE>), dim3(grid), dim3(256), smem, stream, ps, wrows, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo));
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
#endif
  hipLaunchKernelGGL((igemm_win9_kernel<T, MODE>), dim3(grid), dim3(256), smem, stream, p, wrows, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo));
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Persistent kernel for the 64 -> 64 channel 3x3 / stride 1 / pad 1 convolutions (ResNet layer1; MODE 0 forward, MODE 1 data
// gradient), bf16.  These layers have 9 K-steps per tile: with the generic scheme every one of them costs a DMA round trip
// and a barrier, and a tile spends ~12 us on 1 us of MFMA work.  Here the whole weight matrix of a wave's 32 output channels
// (9 taps x 64 channels x 32 = 36 fragments) lives in REGISTERS (144 VGPRs) for the life of the workgroup, so only the
// activations are staged: ONE window of 128 + 2 W + 2 consecutive input pixels (window row j = pixel m0 - W - 1 + j) serves
// all nine taps of a 128-pixel tile — tap (kr, ks) of tile row r reads window row r + W kr + ks; lanes whose tap falls off the
// image (first / last row or column) read a zero row.  The window is double-buffered (the DMA of tile t + 1 runs under tile t),
// the K loop has no barrier and no wait, workgroups are persistent (two per CU) and walk the tile list; the epilogue is
// igemm_epilogue (same fragment layout as the 128 x 64 tiles of the other kernels).
// ---------------------------------------------------------------------------------------------------------
constexpr int kC64WinRows = 248;                 // 128 + 2 * 56 + 2 = 242 rows, rounded up to the 8-row DMA piece
constexpr int kC64WinBytes = kC64WinRows * 128;
constexpr int kC64LoadTap = 4;                   // K-loop tap in front of which the epilogue's residual / mask loads are issued
constexpr int kC64RegTaps = 7;                   // taps whose weight fragments live in registers; the rest are read from LDS
constexpr int kC64Smem = 2 * kC64WinBytes + 256 + 512 + 256 + (9 - kC64RegTaps) * 8192;  // windows, zero pair, column-sum scratch, bias, LDS taps

// HAS_RES / HAS_MSK: whether a residual / a ReLU-mask operand exists is a TEMPLATE parameter — their 16 + 16 registers (loaded in the
// middle of the K loop for the epilogue) otherwise stay reserved in every instance, and with all 256 VGPRs taken the compiler
// serialises the second half of the K loop into read / wait / two MFMAs
template <int MODE, bool HAS_RES, bool HAS_MSK>
__global__ __launch_bounds__(256, 2) void conv64_kernel(const IgemmParams p, const int n_tiles, const FastDiv d_wo, const FastDiv d_howo) {
  using T = bf16raw;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sZ = smem + 2 * kC64WinBytes;
  // sZ: 256 bytes of zeros at a 256-byte boundary = every LDS bank once.  A lane whose tap leaves the image reads the zeros at its
  // OWN position modulo 256, so the 16 lanes of a ds_read_b128 group still hit 16 different bank quads (one shared zero-row
  // address made every edge lane collide with a neighbour in each of its four lane groups: SQ_LDS_BANK_CONFLICT 11-15 %)
  float* sScratch = reinterpret_cast<float*>(sZ + 256);       // [2 wave rows][64] partial column sums
  float* sBias = reinterpret_cast<float*>(sZ + 256 + 512);
  unsigned char* sWt = sZ + 256 + 512 + 256;  // [9 - kC64RegTaps][64 channels][128 B], chunk c of row r stored at c ^ (r & 7)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, g = lane >> 4;
  const int W = p.wo, H = p.ho;
  constexpr int g_first = 0;
  const int g_tiles = n_tiles;
  const int g_blocks = (int)gridDim.x;
  const int g_bid = (int)blockIdx.x;
  const unsigned char* w_src = reinterpret_cast<const unsigned char*>(p.wt);
  const float* bias_src = p.bias;
  if (tid < 16) reinterpret_cast<uint4*>(sZ)[tid] = make_uint4(0, 0, 0, 0);
  if (tid >= 64 && tid < 128) sBias[tid - 64] = bias_src ? bias_src[tid - 64] : 0.f;
  // output / residual / mask / column-sum buffers as buffer resources: rows past M get an out-of-range offset, so every
  // lane issues the same instructions whatever its rows are (loads return 0, stores are dropped)
  const int io_bytes = (int)((long long)p.M * p.ldo * 2);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, io_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.resid), 0, p.resid ? io_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.mask), 0, p.mask ? io_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_cs = __builtin_amdgcn_make_buffer_rsrc(p.colsum_part, 0, p.colsum_part ? (int)((long long)n_tiles * p.ldo * 4) : 0, 0x00020000);
  // The stores of a tile are HELD in registers and issued at the top of the next tile, behind that tile's barrier and window
  // DMA: the wait in front of the barrier then never sees a young store (vmcnt counts stores too), and the stores drain under
  // the next tile's MFMAs.  (With the stores issued in the epilogue the loop spent more than half its time on that wait.)
  u32x4 held[4] = {};
  uint32_t held_off[4] = {kOob, kOob, kOob, kOob};
  float held_cs = 0.f;
  uint32_t held_cs_off = kOob;

  // ---- weights -> registers: fragment j (of 2) of tap t, K half h = 16 bytes of weight row wc*32 + (i16>>2)*8 + j*4 + (i16&3)
  // (the permuted row order the epilogue expects, CPL = 8) at K offset t*64 + h*32 + g*8 ----
  // (all nine taps = 144 VGPRs made the compiler spill inside the K loop, and every reload waits vmcnt(0) = for the stores and
  // the next window; the last taps therefore stay in LDS: 4 extra fragment reads per tap and tile)
  u32x4 fw[kC64RegTaps][2][2];
  const int wrow0 = wc * 32 + (i16 >> 2) * 8 + (i16 & 3);  // + j*4: this lane's weight rows
  for (int i = tid; i < (9 - kC64RegTaps) * 512; i += 256) {
    const int tt = i >> 9, r = (i >> 3) & 63, c = i & 7;
    *reinterpret_cast<uint4*>(sWt + tt * 8192 + r * 128 + ((c ^ (r & 7)) << 4)) =
        *reinterpret_cast<const uint4*>(w_src + (size_t)r * (576 * 2) + (kC64RegTaps + tt) * 128 + c * 16);
  }
  {
    const unsigned char* wbase = w_src + (size_t)wrow0 * (576 * 2) + g * 16;
#pragma unroll
    for (int t = 0; t < kC64RegTaps; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[t][h][j] = *reinterpret_cast<const u32x4*>(wbase + (size_t)j * 4 * (576 * 2) + t * 128 + h * 64);
  }
  // The weight loads are complete HERE, in a form the compiler's wait-count pass sees (it does not read inline asm): without this
  // it keeps one `s_waitcnt vmcnt(N)` per weight fragment inside the tile loop (needed in the first iteration only), and in the
  // steady state those waits stall the K loop on the window DMA of the NEXT tile and on the held stores issued at the tile's top.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched

  const unsigned long long a_ptr = (unsigned long long)p.in;
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * (8 * 128);
  const int lrow = tid >> 3;                       // window row (mod 32) this thread stages
  const int lchunk = (tid & 7) ^ (lrow & 7);       // source-side swizzle: LDS chunk c of window row j holds source chunk c ^ (j & 7)
  constexpr int PSTR = 32 * 128;

  // stage the window of (remapped) tile tl into buffer buf: 8 passes of 32 rows; rows 248.. (wave 3 of the last pass) do not exist
  auto issue_window = [&](int tl, int buf) {
    const int q0 = tl * 128 - W - 1;  // input pixel of window row 0
    uint32_t vo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = q0 + lrow + 32 * i;
      vo[i] = (uint32_t)q * 128u + (uint32_t)(lchunk * 16);  // q < 0 wraps past, q >= M runs past the descriptor's range: zero-filled
    }
    const uint32_t l0 = lds_wave + (uint32_t)(buf * kC64WinBytes);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const uint32_t l_ = l0 + (uint32_t)(2 * q * PSTR);
      asm volatile(
          "s_nop 4\n\t"
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, 0 offen lds\n\t"
          "s_add_u32 m0, %2, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds"
          ::"v"(vo[2 * q]), "v"(vo[2 * q + 1]), "s"(l_), "s"(rs_a), "n"(PSTR)
          : "memory", "scc");
    }
    {
      const uint32_t l_ = l0 + (uint32_t)(6 * PSTR);
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(vo[6]), "s"(l_), "s"(rs_a) : "memory");
      if (wave_u < 3) {
        const uint32_t l2_ = l_ + (uint32_t)PSTR;
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(vo[7]), "s"(l2_), "s"(rs_a) : "memory");
      }
    }
  };

  // Forward (round 5): the next tile's window is issued in four parts between the MFMAs of this tile's K loop (pieces 2 q, 2 q + 1 at
  // half-step 2 + 4 q) instead of eight pieces in a burst at the tile's top: conv64 forward 0.812 -> 0.784 ms per update
  // (profiles/r05h_ab_conv64_spread_dma.txt).  The data-gradient instances keep the burst: with the parts inside their K loop two
  // of them spill six registers there and lose 10 % (0.351 -> 0.387).  -DVDQN_C64_SPREAD=0 / =2: never / both modes (A/B builds).
  auto issue_window_part = [&](int tl, int buf, int part) {
    const int q0 = tl * 128 - W - 1;
    const uint32_t v0 = (uint32_t)(q0 + lrow + 32 * (2 * part)) * 128u + (uint32_t)(lchunk * 16);
    const uint32_t v1 = v0 + 32u * 128u;
    const uint32_t l_ = lds_wave + (uint32_t)(buf * kC64WinBytes) + (uint32_t)(2 * part * PSTR);
    if (part < 3) {
      asm volatile(
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, 0 offen lds\n\t"
          "s_add_u32 m0, %2, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds"
          ::"v"(v0), "v"(v1), "s"(l_), "s"(rs_a), "n"(PSTR)
          : "memory", "scc");
    } else {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(v0), "s"(l_), "s"(rs_a) : "memory");
      if (wave_u < 3) {
        const uint32_t l2_ = l_ + (uint32_t)PSTR;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(v1), "s"(l2_), "s"(rs_a) : "memory");
      }
    }
  };
  (void)issue_window_part;
#ifndef VDQN_C64_SPREAD
#define VDQN_C64_SPREAD 1
#endif
  constexpr bool kSpread = VDQN_C64_SPREAD == 2 || (VDQN_C64_SPREAD == 1 && MODE == 0);
  const uint32_t c64_adv_oh = (uint32_t)((16 / W) % H), c64_adv_ow = (uint32_t)(16 - (16 / W) * W);  // (whole images drop out of the position)
  int t = g_bid, buf = 0;
  if (t < g_tiles) issue_window(g_first + (int)xcd_remap((uint32_t)t, (uint32_t)g_tiles), 0);
  for (; t < g_tiles; t += g_blocks, buf ^= 1) {
    const int tl = g_first + (int)xcd_remap((uint32_t)t, (uint32_t)g_tiles);
    const int m0 = tl * 128;
#ifdef VDQN_STAMP
    const unsigned long long st_w0 = __builtin_amdgcn_s_memtime();  // (tools/stamp_conv64.py) arrival at the tile's top
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // window t visible; everyone is done with the other buffer (and with the scratch)
#ifdef VDQN_STAMP
    // row of 8 words per (workgroup, tile number < 16): arrival, top (barrier passed), DMA issued, held stores + edge bits done
    // (K loop begins), K loop done, epilogue done
    const int st_k = (t - g_bid) / g_blocks;
    unsigned long long* st_row = (p.pool_out && st_k < 16) ? reinterpret_cast<unsigned long long*>(p.pool_out) + ((size_t)blockIdx.x * 16 + st_k) * 8 : nullptr;
    if (st_row && tid == 0) { st_row[0] = st_w0; st_row[1] = __builtin_amdgcn_s_memtime(); }
#endif
    const bool c64_has_nx = t + g_blocks < g_tiles;
    const int c64_tl_nx = c64_has_nx ? g_first + (int)xcd_remap((uint32_t)(t + g_blocks), (uint32_t)g_tiles) : 0;
    if constexpr (!kSpread) {
      if (c64_has_nx) issue_window(c64_tl_nx, buf ^ 1);
    }
#ifdef VDQN_STAMP
    if (st_row && tid == 0) st_row[2] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int f = 0; f < 4; ++f) __builtin_amdgcn_raw_buffer_store_b128(held[f], r_out, (int)held_off[f], 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(held_cs), r_cs, (int)held_cs_off, 0, 0);

    // edge bits of this lane's four pixels (one nibble each, one register): 1 top row, 2 bottom row, 4 left column, 8 right column
    // (the position of pixel f = 0 by division, the three others 16 pixels on each: adds, compares and selects — round 5)
    uint32_t edge = 0u;
    {
      const uint32_t m = (uint32_t)(m0 + wr * 64 + i16);
      const uint32_t rem = m - fastdiv(m, d_howo) * d_howo.div;
      uint32_t oh = fastdiv(rem, d_wo), ow = rem - oh * d_wo.div;
      const uint32_t adv_oh = c64_adv_oh, adv_ow = c64_adv_ow;  // 16 pixels = adv_oh rows + adv_ow columns
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        edge |= ((oh == 0 ? 1u : 0u) | (oh == (uint32_t)H - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == (uint32_t)W - 1 ? 8u : 0u)) << (4 * f);
        ow += adv_ow;
        oh += adv_oh;
        const bool c1 = ow >= (uint32_t)W;
        ow -= c1 ? (uint32_t)W : 0u;
        oh += c1 ? 1u : 0u;
        oh -= oh >= (uint32_t)H ? (uint32_t)H : 0u;  // (into the next image: at most once, adv_oh < H)
      }
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* a_rd = smem + buf * kC64WinBytes + (wr * 64 + i16) * 128;
    // epilogue operands: offsets now, the residual / mask loads are issued in the middle of the K loop (tap kC64LoadTap) so
    // that the remaining taps' MFMAs cover most of their latency
    const int ncol = wc * 32 + g * 8;
    uint32_t off[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int m = m0 + wr * 64 + f * 16 + i16;
      off[f] = m < p.M ? (uint32_t)(m * p.ldo + ncol) * 2u : kOob;
    }
    u32x4 rv[4], mv[4];
    // 18 half-steps (tap, K half); the forward reads input pixel m + (kr-1) W + (ks-1), the data gradient m + (1-kr) W + (1-ks).
    // Software-pipelined by hand: the fragment reads of half-step hs + 1 are issued among the MFMAs of half-step hs (two register
    // sets, sched_group_barrier pattern) — left to itself the compiler emits read / s_waitcnt lgkmcnt(0) / two MFMAs, i.e. one
    // exposed LDS latency per pair of MFMAs (round 3: ISA inspection).
    // (the instance with BOTH epilogue operands has no registers for a second set — it spills 13 of them, and every reload waits
    // vmcnt(0), i.e. for the next tile's window: it keeps one set and the compiler's order)
    constexpr int NSET = (HAS_RES && HAS_MSK) ? 1 : 2;
    u32x4 fa[NSET][4], wl[NSET][2];
#define VDQN_C64_LOAD(HS, SET)                                                                                              \
  {                                                                                                                         \
    constexpr int tap_ = (HS) >> 1, h_ = (HS) & 1;                                                                          \
    constexpr int kr_ = tap_ / 3, ks_ = tap_ % 3;                                                                           \
    constexpr int ky_ = MODE == 0 ? kr_ : 2 - kr_, kx_ = MODE == 0 ? ks_ : 2 - ks_; /* window offsets */                    \
    constexpr uint32_t tapbits_ = (ky_ == 0 ? 1u : 0u) | (ky_ == 2 ? 2u : 0u) | (kx_ == 0 ? 4u : 0u) | (kx_ == 2 ? 8u : 0u); \
    const int off_ = W * ky_ + kx_; /* wave-uniform window row offset of this tap */                                        \
    const int key_ = (i16 + off_) & 7;                                                                                      \
    const int par_ = ((i16 + off_) & 1) << 7; /* (window buffers start at multiples of 256 bytes: row parity = address bit 7) */ \
    const int coff_ = ((g + 4 * h_) ^ key_) << 4;                                                                           \
    const unsigned char* z_rd_ = sZ + par_ + coff_; /* the lane's own position inside the zero pair */                      \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                                      \
      const bool z_ = (edge & (tapbits_ << (4 * f_))) != 0u;                                                                \
      fa[SET][f_] = *reinterpret_cast<const u32x4*>(z_ ? z_rd_ : a_rd + (f_ * 16 + off_) * 128 + coff_);                    \
    }                                                                                                                       \
    if constexpr (tap_ >= kC64RegTaps) {                                                                                    \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                      \
          wl[SET][j_] = *reinterpret_cast<const u32x4*>(sWt + (tap_ - kC64RegTaps) * 8192 + (wrow0 + j_ * 4) * 128 +        \
                                                        (((g + 4 * h_) ^ ((wrow0 + j_ * 4) & 7)) << 4));                    \
    }                                                                                                                       \
  }
#define VDQN_C64_MMA(HS, SET)                                                                                               \
  {                                                                                                                         \
    constexpr int tap_ = (HS) >> 1, h_ = (HS) & 1;                                                                          \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                     \
      u32x4 w_;                                                                                                             \
      if constexpr (tap_ < kC64RegTaps) w_ = fw[tap_ < kC64RegTaps ? tap_ : 0][h_][j_];                                     \
      else w_ = wl[SET][j_];                                                                                                \
      acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_), __builtin_bit_cast(bf16x8, fa[SET][f_]), acc[f_][j_], 0, 0, 0); \
    }                                                                                                                       \
  }
    // one pipelined half-step: reads of HS + 1 (4, or 6 with LDS weights) spread over the 8 MFMAs of HS
#define VDQN_C64_SPREAD_ISSUE(HS)                                                                                           \
    if constexpr (kSpread && ((HS)&3) == 2 && (HS) < 16) {                                                                  \
      if (c64_has_nx) issue_window_part(c64_tl_nx, buf ^ 1, (HS) >> 2);                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                                    \
    }
#define VDQN_C64_STEP(HS)                                                                                                   \
  {                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    VDQN_C64_SPREAD_ISSUE(HS)                                                                                               \
    if constexpr ((HS) == 2 * kC64LoadTap) {                                                                                \
      if constexpr (HAS_RES) {                                                                                              \
        _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) rv[f_] = __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)off[f_], 0, 0); \
      }                                                                                                                     \
      if constexpr (HAS_MSK) {                                                                                              \
        _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) mv[f_] = __builtin_amdgcn_raw_buffer_load_b128(r_msk, (int)off[f_], 0, 0); \
      }                                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                                    \
    }                                                                                                                       \
    if constexpr (NSET == 2) {                                                                                              \
      if constexpr ((HS) + 1 < 18) VDQN_C64_LOAD((HS) + 1, ((HS) + 1) & 1)                                                  \
      VDQN_C64_MMA(HS, (HS) & 1)                                                                                            \
      _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) {                                                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                  \
      }                                                                                                                     \
    } else {                                                                                                                \
      if constexpr ((HS) > 0) VDQN_C64_LOAD(HS, 0)                                                                          \
      VDQN_C64_MMA(HS, 0)                                                                                                   \
    }                                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
  }
#ifdef VDQN_STAMP
    if (st_row && tid == 0) st_row[3] = __builtin_amdgcn_s_memtime();
#endif
    VDQN_C64_LOAD(0, 0)
    VDQN_C64_STEP(0) VDQN_C64_STEP(1) VDQN_C64_STEP(2) VDQN_C64_STEP(3) VDQN_C64_STEP(4) VDQN_C64_STEP(5)
    VDQN_C64_STEP(6) VDQN_C64_STEP(7) VDQN_C64_STEP(8) VDQN_C64_STEP(9) VDQN_C64_STEP(10) VDQN_C64_STEP(11)
    VDQN_C64_STEP(12) VDQN_C64_STEP(13) VDQN_C64_STEP(14) VDQN_C64_STEP(15) VDQN_C64_STEP(16) VDQN_C64_STEP(17)
#undef VDQN_C64_STEP
#undef VDQN_C64_SPREAD_ISSUE
#undef VDQN_C64_MMA
#undef VDQN_C64_LOAD
#ifdef VDQN_STAMP
    if (st_row && tid == 0) { asm volatile("s_nop 0" ::"v"(acc[3][1][3])); st_row[4] = __builtin_amdgcn_s_memtime(); }
#endif
    // ---- epilogue (the arithmetic of igemm_epilogue, CPL = 8): lane (i16, g) owns channels ncol .. ncol + 7 of pixels f*16 + i16 ----
    {
      float bv[8];
      *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(sBias + ncol);
      *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(sBias + ncol + 4);
      float cs[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[f][j][r] + bv[j * 4 + r];
        if constexpr (HAS_RES) {
          const bf16raw* pr = reinterpret_cast<const bf16raw*>(&rv[f]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32(pr[e]);
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (HAS_MSK) {
          const bf16raw* pm = reinterpret_cast<const bf16raw*>(&mv[f]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16_to_f32(pm[e]) > 0.f) ? v[e] : 0.f;
        }
        bf16raw ov[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = f32_to_bf16(v[e]);
        if constexpr (MODE == 1) {  // column sums exist for the data gradient only (the dispatch keeps a forward call that asks for them off this kernel)
          const float live = off[f] != kOob ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[e] += live * bf16_to_f32(ov[e]);
        }
        held[f] = *reinterpret_cast<const u32x4*>(ov);
        held_off[f] = off[f];
      }
      if (MODE == 1 && p.colsum_part) {  // uniform: partial column sums of this tile -> colsum_part[tile][ldo]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t_ = cs[e];
          t_ += __shfl_xor(t_, 1, 64);
          t_ += __shfl_xor(t_, 2, 64);
          t_ += __shfl_xor(t_, 4, 64);
          t_ += __shfl_xor(t_, 8, 64);
          cs[e] = t_;
        }
        // (the scratch was last read before this tile's top barrier)
        if (i16 == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) sScratch[wr * 64 + ncol + e] = cs[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 64) {
          held_cs = sScratch[tid] + sScratch[64 + tid];
          held_cs_off = (uint32_t)(tl * p.ldo + tid) * 4u;
        }
      }
    }
#ifdef VDQN_STAMP
    if (st_row && tid == 0) st_row[5] = __builtin_amdgcn_s_memtime();
#endif
  }
  // the last tile's stores
#pragma unroll
  for (int f = 0; f < 4; ++f) __builtin_amdgcn_raw_buffer_store_b128(held[f], r_out, (int)held_off[f], 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(held_cs), r_cs, (int)held_cs_off, 0, 0);
}

template <int MODE>
int launch_conv64(const IgemmParams& p, hipStream_t stream) {
  const bool has_res = p.resid != nullptr, has_msk = p.mask != nullptr;
  auto kern = has_res ? (has_msk ? &conv64_kernel<MODE, true, true> : &conv64_kernel<MODE, true, false>)
                      : (has_msk ? &conv64_kernel<MODE, false, true> : &conv64_kernel<MODE, false, false>);
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(kern), (size_t)kC64Smem);
  const int n_cu = vdqn_num_cus();
  const int n_tiles = (p.M + 127) / 128;
  int grid = n_tiles < 2 * n_cu ? n_tiles : 2 * n_cu;
  vdqn_prof_begin(MODE == 0 ? "conv64<bf16,fwd>" : "conv64<bf16,dgrad>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))), stream);
#ifdef VDQN_STAMP
  IgemmParams ps = p;
  ps.pool_out = g_stamp_buffer;  // (conv64 has no pooled output: the field carries the stamp buffer in diagnostic builds)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kC64Smem, stream, ps, n_tiles, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo));
#else
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kC64Smem, stream, p, n_tiles, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo));
#endif
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

template <typename T, int BM, int BN, int MODE, int NS = 2>
int launch_igemm(const IgemmParams& p, hipStream_t stream) {
  // 256x64 tiles: 4 waves of 64x64 (WN = 1); everything else a (BM/64) x 2 wave grid
  constexpr int WN = (BM == 256 && BN == 64) ? 1 : 2;
  const size_t smem = NS * (BM + BN) * 128;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, MODE, WN, NS>), (size_t)smem);
  const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
  const double esz = sizeof(T);
  // one tag per kernel symbol (T, BM, BN, MODE), so bench.py rows line up with rocprofv3's kernel names
  static const char* const kTag[2][4][3] = {{{"igemm<bf16,64,fwd>", "igemm<bf16,64,dgrad>", "igemm<bf16,64,dgrad_s2>"},
                                             {"igemm<bf16,128,fwd>", "igemm<bf16,128,dgrad>", "igemm<bf16,128,dgrad_s2>"},
                                             {"igemm<bf16,256x64,fwd>", "igemm<bf16,256x64,dgrad>", "igemm<bf16,256x64,dgrad_s2>"},
                                             {"igemm<bf16,256x128,fwd>", "igemm<bf16,256x128,dgrad>", "igemm<bf16,256x128,dgrad_s2>"}},
                                            {{"igemm<f32,64,fwd>", "igemm<f32,64,dgrad>", "igemm<f32,64,dgrad_s2>"},
                                             {"igemm<f32,128,fwd>", "igemm<f32,128,dgrad>", "igemm<f32,128,dgrad_s2>"},
                                             {"igemm<f32,256x64,fwd>", "igemm<f32,256x64,dgrad>", "igemm<f32,256x64,dgrad_s2>"},
                                             {"igemm<f32,256x128,fwd>", "igemm<f32,256x128,dgrad>", "igemm<f32,256x128,dgrad_s2>"}}};
  if constexpr (MODE == 3)
    vdqn_prof_begin(sizeof(T) == 2 ? "stem_conv_pool<bf16>" : "stem_conv_pool<f32>", 2.0 * p.n_img * 112 * 112 * 64 * 147,
                    esz * ((double)p.n_img * 115 * 115 * 16 + 64.0 * 256 + (double)p.n_img * 56 * 56 * 64) + (double)p.n_img * 56 * 56 * 64, stream);
  else
    vdqn_prof_begin(kTag[sizeof(T) == 2 ? 0 : 1][BM == 256 ? (BN == 128 ? 3 : 2) : (BN == 128 ? 1 : 0)][MODE],
                    2.0 * p.M * p.co * p.ktot,
                    esz * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))),
                    stream);
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, MODE, WN, NS>), dim3(grid), dim3(BM * WN), smem, stream, p);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

template <typename T, int BM, int BN>
int launch_mode(const IgemmParams& p, int mode, hipStream_t st) {
  if constexpr (BM == 128) {
    // at most one tile per CU and a long K loop: nothing but a deeper ring hides the DMA round trip of every K-step
    static const bool deep_on = !getenv("VDQN_IGEMM_DEEP") || atoi(getenv("VDQN_IGEMM_DEEP")) != 0;
    if (mode == 0 && deep_on && p.tiles_m * p.tiles_n <= 256 && p.nk >= 8) return launch_igemm<T, BM, BN, 0, 4>(p, st);
  }
  if (mode == 0) return launch_igemm<T, BM, BN, 0>(p, st);
  if (mode == 1) return launch_igemm<T, BM, BN, 1>(p, st);
  if constexpr (BM == 128) return launch_igemm<T, BM, BN, 2>(p, st);
  return VDQN_ERR_INVALID;
}

}  // namespace

#ifdef VDQN_STAMP
extern "C" void vdqn_debug_stamp_buffer(void* p) { g_stamp_buffer = p; }  // diagnostic builds only (not part of include/vdqn.h)
#endif

extern "C" int vdqn_conv2d(const vdqn_conv_args* a, void* stream) {
  VDQN_CHECK(a != nullptr, "vdqn_conv2d: null args");
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
  p.pool_out = nullptr; p.pool_idx = nullptr;
  p.in2 = nullptr; p.wt2 = nullptr; p.bias2 = nullptr; p.out2 = nullptr;
  p.co2 = p.ldo2 = p.relu2 = p.ci2 = p.wt2_bytes = 0;
  static const int no_lean = [] { const char* e = getenv("VDQN_LEAN_EPILOGUE"); return (e && e[0] == '0') ? 1 : 0; }();
  p.no_lean = no_lean;
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
  p.cls_interleave = 0;
  if (a->mode == 1 && a->stride == 2) {  // tiles are laid out parity class by parity class (see the kernel)
    p.cls_h[0] = (a->ho + 1) / 2; p.cls_h[1] = a->ho / 2;
    p.cls_w[0] = (a->wo + 1) / 2; p.cls_w[1] = a->wo / 2;
    for (int c = 0; c < 4; ++c) p.cls_tile0[c + 1] = p.cls_tile0[c] + (a->n_img * p.cls_h[c >> 1] * p.cls_w[c & 1] + 127) / 128;
    p.tiles_m = p.cls_tile0[4];
    static const int inter = [] { const char* e = getenv("VDQN_DGRAD_S2_INTERLEAVE"); return e ? atoi(e) : 1; }();
    p.cls_interleave = inter && p.cls_h[0] == p.cls_h[1] && p.cls_w[0] == p.cls_w[1];
  }
  p.tiles_n = (a->co + bn - 1) / bn;
  p.tiles_n1 = p.tiles_n;  // no sibling tiles unless set below
  const bool has_sib = a->wt2 != nullptr;
  if (has_sib) {
    VDQN_CHECK(a->r == 3 && a->s == 3 && a->stride == 2 && a->pad == 1, "vdqn_conv2d: a sibling 1x1 needs a 3x3 / stride-2 / pad-1 call");
    VDQN_CHECK((((uintptr_t)a->wt2) & 15) == 0, "vdqn_conv2d: wt2 must be 16-byte aligned");
    if (a->mode == 0) {
      VDQN_CHECK(a->out2 && a->co2 > 0 && a->ldo2 >= a->co2 && a->co2 % bn == 0 && bn == 128 && !a->out_f32 && !a->colsum_part,
                 "vdqn_conv2d: fused sibling (forward) needs out2, co2 a multiple of the 128-column tile, no f32 copy / column sums");
      VDQN_CHECK(((a->ldo2 * esz) % 16 == 0) && ((((uintptr_t)a->out2) & 15) == 0) && (a->ldo2 % 8 == 0), "vdqn_conv2d: out2 must be 16-byte aligned rows");
      p.wt2 = a->wt2; p.bias2 = a->bias2; p.out2 = a->out2; p.co2 = a->co2; p.ldo2 = a->ldo2; p.relu2 = a->relu2;
      p.wt2_bytes = (int)((long long)a->co2 * a->ci * esz);
      p.tiles_n = p.tiles_n1 + a->co2 / bn;
    } else {
      VDQN_CHECK(a->in2 && a->ci2 > 0 && a->ci2 % kc == 0 && a->ci2 <= a->pix_stride && (((uintptr_t)a->in2) & 15) == 0,
                 "vdqn_conv2d: fused sibling (data gradient) needs in2 with ci2 (a multiple of %d) channels in pixels of pix_stride elems", kc);
      p.in2 = a->in2; p.wt2 = a->wt2; p.ci2 = a->ci2;
      p.wt2_bytes = (int)((long long)p.tiles_n * bn * a->ci2 * esz);
    }
  }
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
  // the Q-head's skinny GEMMs (bf16 linear layers forward / data gradient, features.8 forward): small tiles, K split over the waves
  if (mode != 2) {
    const int sk = vdqn_skinny_kind(a);
    if (sk) return vdqn_launch_skinny(&p, sk, st);
  }
  // 64-column layers with many rows: 256-row tiles, 8 waves (more MFMA work per DMA round trip, half the weight traffic)
  // (VDQN_BM256_MIN_ROWS overrides the row threshold: tests lower it to reach this variant with small tensors, a huge value disables it)
  // 3x3 / stride 1 / pad 1: the window kernel (one staged activation window per kernel row and channel chunk)
  // 64 -> 64 channels, bf16: the persistent kernel with the weights in registers (VDQN_CONV64=0 falls back to the window kernel)
  static const int use_c64 = [] { const char* e = getenv("VDQN_CONV64"); return e ? atoi(e) : 1; }();
  if (use_c64 && a->dtype == VDQN_BF16 && a->r == 3 && a->s == 3 && a->stride == 1 && a->pad == 1 && mode != 2 && a->ci == 64 && a->co == 64 &&
      a->pix_stride == 64 && a->hi == a->ho && a->wi == a->wo && a->wo >= 2 && a->wo <= 56 && p.in_bytes < 0x7fffffffLL && (long long)p.M * 128 < 0x7fffffffLL &&
      a->out && !a->out_f32 && p.vec_ok && (long long)p.M * a->ldo * 2 < 0x7fffffffLL && (mode == 1 || !a->colsum_part)) {
    return mode == 0 ? launch_conv64<0>(p, st) : launch_conv64<1>(p, st);
  }
  static const int use_win = [] { const char* e = getenv("VDQN_IGEMM_WINDOW"); return e ? atoi(e) : 1; }();
  if (use_win && a->r == 3 && a->s == 3 && a->stride == 1 && a->pad == 1 && mode != 2 && a->pix_stride == a->ci && a->wo >= 2) {
    if (a->dtype == VDQN_BF16) {
      // nine taps per staged window (VDQN_IGEMM_WINDOW=2 keeps the three-tap windows): images up to 28 pixels wide, offsets < 2 GiB
      if (bn == 128 && use_win != 2 && a->wo <= 28 && a->hi == a->ho && a->wi == a->wo && p.in_bytes < 0x7fffffffLL) {
        // VDQN_WIN9_UNROLLED (default 1): the K loop unrolled over a chunk pair (win9.hip: half the instructions per K-step); 0: igemm_win9_kernel
        static const int unrolled = [] { const char* e = getenv("VDQN_WIN9_UNROLLED"); return e ? atoi(e) : 1; }();
        if (unrolled && a->ci % 128 == 0) {
          return vdqn_launch_win9u(&p, mode, st);
        }
        return mode == 0 ? launch_igemm_win9<bf16raw, 0>(p, st) : launch_igemm_win9<bf16raw, 1>(p, st);
      }
      if (bn == 128) return mode == 0 ? launch_igemm_win<bf16raw, 128, 0>(p, st) : launch_igemm_win<bf16raw, 128, 1>(p, st);
      return mode == 0 ? launch_igemm_win<bf16raw, 64, 0>(p, st) : launch_igemm_win<bf16raw, 64, 1>(p, st);
    }
    if (bn == 128) return mode == 0 ? launch_igemm_win<float, 128, 0>(p, st) : launch_igemm_win<float, 128, 1>(p, st);
    return mode == 0 ? launch_igemm_win<float, 64, 0>(p, st) : launch_igemm_win<float, 64, 1>(p, st);
  }
  // 3x3 / stride 2 / pad 1 forward over an even-sized input (conv1 of layer2.0 / layer3.0 / layer4.0), bf16, 128-column tiles: the
  // plane-window kernel (win9s.hip; VDQN_S2WIN=0 keeps the generic kernel).  A fused sibling 1x1 (the block's
  // downsample) rides in the same persistent launch for 1, 2 or 4 channel chunks (round 5).
  static const int use_s2win = [] { const char* e = getenv("VDQN_S2WIN"); return e ? atoi(e) : 1; }();
  if (use_s2win && mode == 0 && a->dtype == VDQN_BF16 && a->r == 3 && a->s == 3 && a->stride == 2 && a->pad == 1 &&
      bn == 128 && a->co % 128 == 0 && a->hi == 2 * a->ho && a->wi == 2 * a->wo && a->wo >= 2 && a->wo <= 28 && a->pix_stride == a->ci && a->ci % 64 == 0 &&
      p.in_bytes < 0x7fffffffLL && !a->colsum_part && !a->mask && vdqn_win9s_supports(a->ci / 64, has_sib) &&
      (!has_sib || (a->co2 == a->co && p.vec_ok && !a->out_f32 && !a->resid))) {
    return vdqn_launch_win9s(&p, st);
  }
  // data gradient of 3x3 / stride 2 / pad 1 over an even-sized image, bf16: the plane-window kernel (win9d.hip, round 5;
  // VDQN_S2DGRAD_WIN=0 keeps the generic parity-class tiles)
  static const int use_s2dwin = [] { const char* e = getenv("VDQN_S2DGRAD_WIN"); return e ? atoi(e) : 1; }();
  if (use_s2dwin && mode == 2 && a->dtype == VDQN_BF16 && a->r == 3 && a->s == 3 && a->pad == 1 && a->ho == 2 * a->hi && a->wo == 2 * a->wi &&
      a->wi >= 2 && a->wi <= 28 && a->pix_stride == a->ci && p.vec_ok && a->out && !a->out_f32 && !a->bias && !p.no_lean &&
      p.in_bytes < 0x7fffffffLL && (long long)p.M * a->ldo * 2 < 0x7fffffffLL && (long long)a->co * p.ktot * 2 < 0x7fffffffLL &&
      vdqn_win9d_supports(a->ci, a->co, has_sib, a->ci2)) {
    return vdqn_launch_win9d(&p, st);
  }
  static const long long min256 = [] { const char* e = getenv("VDQN_BM256_MIN_ROWS"); return e ? atoll(e) : 256ll * 1024; }();
  if (bn == 64 && mode != 2 && p.M >= min256 && !has_sib) {
    p.tiles_m = (p.M + 255) / 256;
    return a->dtype == VDQN_BF16 ? launch_mode<bf16raw, 256, 64>(p, mode, st) : launch_mode<float, 256, 64>(p, mode, st);
  }
  if (a->dtype == VDQN_BF16) return bn == 128 ? launch_mode<bf16raw, 128, 128>(p, mode, st) : launch_mode<bf16raw, 128, 64>(p, mode, st);
  return bn == 128 ? launch_mode<float, 128, 128>(p, mode, st) : launch_mode<float, 128, 64>(p, mode, st);
}

extern "C" int vdqn_stem_conv_pool_n(const void* t_in, const void* wt, const float* bias, void* pool, void* idx, int32_t n_img, int32_t n_idx_img,
                                     int32_t dtype, void* stream) {
  VDQN_CHECK(t_in && wt && bias && pool && n_img > 0 && n_idx_img >= 0 && n_idx_img <= n_img && (idx || n_idx_img == 0), "vdqn_stem_conv_pool: bad args");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_stem_conv_pool: bad dtype %d", dtype);
  VDQN_CHECK((((uintptr_t)t_in | (uintptr_t)wt | (uintptr_t)pool | (uintptr_t)idx) & 15) == 0, "vdqn_stem_conv_pool: tensors must be 16-byte aligned");
  const int esz = dtype == VDQN_BF16 ? 2 : 4;
  VDQN_CHECK((long long)n_img * 64 < (1ll << 31) / 256, "vdqn_stem_conv_pool: too many images");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = t_in; p.wt = wt; p.bias = bias; p.pool_out = pool; p.pool_idx = (uint8_t*)idx; p.n_idx_img = n_idx_img;
  p.n_img = n_img; p.hi = 115; p.wi = 115; p.ci = 64; p.pix_stride = 16;  // one K-step = 4 packed pixels x 16 = one kernel row
  p.ho = 112; p.wo = 112; p.co = 64; p.ldo = 64; p.r = 4; p.s = 1; p.stride = 1; p.pad = 0; p.relu = 1;
  p.howo = 112 * 112;
  p.ktot = 256;
  p.nk = p.ktot / (128 / esz);
  p.tiles_m = n_img * 64;  // 8 x 8 patches of 7 x 7 pooled pixels per image
  p.tiles_n = 1;
  p.M = p.tiles_m * 256;
  p.in_bytes = (long long)n_img * 115 * 115 * 16 * esz;
  p.wt_bytes = 64 * p.ktot * esz;
  p.vec_ok = 1;
  hipStream_t st = (hipStream_t)stream;
  // bf16: the persistent kernel of stem.hip (weights in registers, double-buffered windows; 0.62 vs 0.70 ms per step for the
  // one-tile-per-workgroup MODE 3 path below, which f32 uses and VDQN_STEM_PERSISTENT=0 selects)
  static const bool persistent = [] { const char* e = getenv("VDQN_STEM_PERSISTENT"); return !(e && e[0] == '0'); }();
  if (dtype == VDQN_BF16 && persistent) return vdqn_stem_bf16(t_in, wt, bias, pool, idx, n_img, n_idx_img, st);
  return dtype == VDQN_BF16 ? launch_igemm<bf16raw, 256, 64, 3>(p, st) : launch_igemm<float, 256, 64, 3>(p, st);
}

extern "C" int vdqn_stem_conv_pool(const void* t_in, const void* wt, const float* bias, void* pool, void* idx, int32_t n_img, int32_t dtype,
                                   void* stream) {
  return vdqn_stem_conv_pool_n(t_in, wt, bias, pool, idx, n_img, idx ? n_img : 0, dtype, stream);
}
