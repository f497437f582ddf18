// Plane-window kernel for the DATA GRADIENT of the 3x3 / stride-2 / pad-1 convolutions (bf16): conv1 of the first BasicBlock of ResNet
// layer2, layer3 and layer4 (torchvision resnet.py, reached from archs/HabitatDQNMultiAction.py:30; the backward of
// train_q_network.py:226), with the data gradient of the block's 1x1 / stride-2 downsample accumulated in the same tiles (SIB).
//
// gx[img, 2y + ph, 2x + pw, :] of output-parity class (ph, pw) only receives the taps with kr = ph + 1, ks = pw + 1 (mod 2), and
// those read gy one pixel down / right or not at all:
//     class (1,1): taps (0,0) (0,2) (2,0) (2,2) at (dy, dx) = (1,1) (1,0) (0,1) (0,0)       class (1,0): (0,1) (2,1) at (1,0) (0,0)
//     class (0,1): taps (1,0) (1,2) at (0,1) (0,0)                                          class (0,0): (1,1) at (0,0)  [+ the 1x1]
// So the four classes are four stride-ONE convolutions over the gy image, and — as in win9s.hip for the forward — ONE staged window
// of 128 + Wo + 2 consecutive gy pixels per 64-channel chunk serves every tap of a class (tile row r reads window row
// r + dy Wo + dx; a lane whose tap leaves the image — bottom row with dy = 1, right column with dx = 1 — reads the zero pair at its
// own bank position).  The generic kernel (igemm_kernel MODE 2) ran each class as its own tiles of 2-8 K-steps, one workgroup per
// tile, every tap re-staging its 128 gy rows: 440 / 318 TFLOP/s with 1.2-1.56x the algorithmic HBM traffic
// (profiles/r04bf_pmc_traffic.json).  Here a workgroup is persistent, a tile is 128 gy pixels x ALL FOUR classes run as four
// accumulation phases (epilogue behind each; the downsample's K-steps extend the last one), the first two K-steps of what comes
// next are always staged under the last two of what runs, and the classes of a pixel block leave the gy rows in L2 for each other.
//
// K order inside a class: (chunk, tap), tap fastest — not the generic kernel's (tap, chunk): equal up to the rounding of another
// summation order.  Column sums: one entry per (tile, class) = 4 x tiles entries, as the generic kernel's class tiles wrote them.
#include <stdlib.h>

#include "igemm_common.h"

namespace {

constexpr int kD_WinRows = 160;  // >= 128 + 28 + 2 (+ the zero pair), a multiple of the 32-row staging pass
constexpr int kD_WinStride = kD_WinRows * 128;
constexpr int kD_WPass = kD_WinRows / 32;

// phase kinds: 0..3 = output classes (1,1) (1,0) (0,1) (0,0); 4 = the sibling 1x1 (accumulates into class (0,0))
struct DTap { int tap, dy, dx; };
__host__ __device__ constexpr int d_len(int kind) { return kind == 0 ? 4 : (kind <= 2 ? 2 : 1); }
__host__ __device__ constexpr DTap d_tap(int kind, int i) {
  constexpr DTap t0[4] = {{0, 1, 1}, {2, 1, 0}, {6, 0, 1}, {8, 0, 0}};
  constexpr DTap t1[2] = {{1, 1, 0}, {7, 0, 0}};
  constexpr DTap t2[2] = {{3, 0, 1}, {5, 0, 0}};
  return kind == 0 ? t0[i & 3] : kind == 1 ? t1[i & 1] : kind == 2 ? t2[i & 1] : kind == 3 ? DTap{4, 0, 0} : DTap{0, 0, 0};
}

template <int NF>
struct DGeom {
  static constexpr int BN = 32 * NF;
  static constexpr int WtTile = BN * 128;
  static constexpr int WinBase = 2 * WtTile;
  static constexpr int Scratch = WinBase + 2 * kD_WinStride;
  static constexpr int Smem = Scratch + 2 * BN * 4;
};

// SPLIT: a workgroup runs ONE group of classes for every tile of its walk — A = class (1,1) (4 taps), B = classes (1,0) and (0,1)
// (2 + 2 taps), C = class (0,0) (+ the 1x1) — and the workgroups of an XCD are divided n_a : n_a : rest between the groups (the
// launcher balances K-steps).  The K-steps of this kernel are bound by the DMA round trip, not by MFMA work, so a launch is as long
// as the longest chain of steps one workgroup runs: layer4.0's 196 tiles x 80 steps (one workgroup each, every second slot of
// the chip empty) become 3 x 196 chains of 32 / 32 / 16 steps.
template <int NF, bool SIB, bool SPLIT>
__global__ __launch_bounds__(256, 2) void win9d_kernel(const IgemmParams p, const FastDiv d_wo, const FastDiv d_howo, const uint32_t total_tiles, const int tiles_n,
                                                       const uint32_t n_a) {
  using T = bf16raw;
  using G = DGeom<NF>;
  constexpr int BM = 128, BN = G::BN, WN = 2, CPL = 4 * NF;
  constexpr int PSTR = 32 * 128;
  constexpr int BPASS = BN / 32;  // weight staging pieces per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tiles of this workgroup: win9u_kernel's XCD-contiguous walk over (row block, column tile), column fastest
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t tq = total_tiles >> 3, tr = total_tiles & 7u;
  const uint32_t x_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const uint32_t x_count = tq + (xcd < tr ? 1u : 0u);
  const uint32_t x_blocks = (gridDim.x >> 3) + (xcd < (gridDim.x & 7u) ? 1u : 0u);
  uint32_t lt = blockIdx.x >> 3, lt_step = x_blocks;
  int grp = 0;  // SPLIT: 0 / 1 / 2 = groups A / B / C
  if constexpr (SPLIT) {  // (the launcher makes the grid a multiple of 8 and n_a >= 1, x_blocks - 2 n_a >= 1)
    grp = lt < n_a ? 0 : (lt < 2u * n_a ? 1 : 2);
    grp = __builtin_amdgcn_readfirstlane(grp);
    lt_step = grp == 2 ? x_blocks - 2u * n_a : n_a;
    lt -= (uint32_t)grp * n_a;
  }
  if (lt >= x_count) return;
  int tile_n = (int)((x_first + lt) % (uint32_t)tiles_n), tile_m = (int)((x_first + lt) / (uint32_t)tiles_n);
  int n0 = tile_n * BN, m0 = tile_m * BM;
  // gy: [n_img][Ho][Wo][ci] (p.hi x p.wi); gx: [n_img][2 Ho][2 Wo][co] (p.ho x p.wo)
  const int Wo = p.wi, Ho = p.hi, Wi = p.wo;
  const int rows_total = p.n_img * Ho * Wo;  // gy pixels = rows of every class
  const int lrow = tid >> 3;
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int lchunk_b = (tid & 7) ^ ((((lrow / CPL) & 1) << 2) | (lrow & 3));
  const int cpk = p.ci / 64;           // channel chunks of gy (even: the dispatch checks)
  const int tap_k = p.ci * 2;          // bytes between the weight K offsets of consecutive taps of one chunk
  const int b_row32 = 32 * 9 * tap_k;  // 32 weight rows of the 3x3 ([co][3][3][ci] bf16)
  const int b2_row32 = 32 * tap_k;     // ... of the 1x1 ([co][ci2], ci2 == ci)
  const int pixB = tap_k;              // bytes per gy pixel (pix_stride == ci)
  const int need = BM + Wo + 2;

  const unsigned long long a_ptr = (unsigned long long)p.in, b_ptr = (unsigned long long)p.wt;
  const unsigned long long a2_ptr = (unsigned long long)(SIB ? p.in2 : p.in), b2_ptr = (unsigned long long)(SIB ? p.wt2 : p.wt);
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const i32x4 rs_a2 = {__builtin_amdgcn_readfirstlane((int)(unsigned)a2_ptr), __builtin_amdgcn_readfirstlane((int)((a2_ptr >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};
  const i32x4 rs_b2 = {__builtin_amdgcn_readfirstlane((int)(unsigned)b2_ptr), __builtin_amdgcn_readfirstlane((int)((b2_ptr >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane(SIB ? p.wt2_bytes : p.wt_bytes), 0x00020000};

  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * (8 * 128);
  const uint32_t a_lane = (uint32_t)(lchunk_a * 16);
  // per-lane weight row offsets WITHOUT the column tile: n0's rows go into the DMA's scalar offset (b_n0 / b2_n0), so a tile switch
  // changes scalars only and the next tile's offsets need no registers
  const uint32_t b_off0 = (uint32_t)lrow * (uint32_t)(9 * tap_k) + (uint32_t)(lchunk_b * 16);
  const uint32_t b2_off0 = (uint32_t)lrow * (uint32_t)tap_k + (uint32_t)(lchunk_b * 16);
  int b_n0 = n0 * 9 * tap_k, b2_n0 = n0 * tap_k;

  // window of gy pixels Q0_ .. (descriptor RSA_, chunk / tensor offset SO_) -> window buffer WBUF: five pieces.  The per-lane offsets
  // are LINEAR in the pixel index (gy is dense): rebuilt here from one base instead of held in registers
#define VDQN_D_ISSUE_AW(WBUF, Q0_, RSA_, SO_)                                                                      \
  {                                                                                                                \
    const uint32_t la_ = lds_wave + (uint32_t)(G::WinBase + (WBUF)*kD_WinStride);                                  \
    const i32x4 rsw_ = (RSA_);                                                                                     \
    const int so_ = (SO_);                                                                                         \
    int q_ = (Q0_) + lrow;                                                                                         \
    asm volatile("" : "+v"(q_));                                                                                   \
    uint32_t ar_[kD_WPass];                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < kD_WPass; ++i_) {                                                      \
      const int qi_ = q_ + 32 * i_;                                                                                \
      ar_[i_] = (lrow + 32 * i_ < need && (unsigned)qi_ < (unsigned)rows_total) ? (uint32_t)qi_ * (uint32_t)pixB + a_lane : kOob; \
    }                                                                                                              \
    asm volatile(                                                                                                  \
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %6, %7 offen lds\n\t"                             \
        "s_add_u32 m0, %5, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %6, %7 offen lds\n\t"                         \
        "s_add_u32 m0, %5, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %7 offen lds\n\t"                         \
        "s_add_u32 m0, %5, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %7 offen lds\n\t"                        \
        "s_add_u32 m0, %5, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %7 offen lds"                             \
        ::"v"(ar_[0]), "v"(ar_[1]), "v"(ar_[2]), "v"(ar_[3]), "v"(ar_[4]), "s"(la_), "s"(rsw_), "s"(so_),          \
          "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR), "n"(4 * PSTR)                                                   \
        : "memory", "scc");                                                                                        \
  }
  // weight tile -> weight buffer BUF: BPASS pieces (rows lrow + 32 i of the column tile at per-lane offset VOFF_ of descriptor RS_,
  // K offset SO0_, RSTR_ bytes per 32 rows)
#define VDQN_D_ISSUE_B(BUF, VOFF_, RS_, SO0_, RSTR_)                                                               \
  {                                                                                                                \
    const uint32_t lb_ = lds_wave + (uint32_t)((BUF)*G::WtTile);                                                   \
    const i32x4 rsb_ = (RS_);                                                                                      \
    const int so0_ = (SO0_), so1_ = so0_ + (RSTR_);                                                                \
    if constexpr (BPASS == 4) {                                                                                    \
      const int so2_ = so1_ + (RSTR_), so3_ = so2_ + (RSTR_);                                                      \
      asm volatile(                                                                                                \
          "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\t"                           \
          "s_add_u32 m0, %1, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %4 offen lds\n\t"                       \
          "s_add_u32 m0, %1, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %5 offen lds\n\t"                       \
          "s_add_u32 m0, %1, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %6 offen lds"                            \
          ::"v"(VOFF_), "s"(lb_), "s"(rsb_), "s"(so0_), "s"(so1_), "s"(so2_), "s"(so3_), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
          : "memory", "scc");                                                                                      \
    } else {                                                                                                       \
      asm volatile(                                                                                                \
          "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\t"                           \
          "s_add_u32 m0, %1, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %4 offen lds"                            \
          ::"v"(VOFF_), "s"(lb_), "s"(rsb_), "s"(so0_), "s"(so1_), "n"(PSTR)                                       \
          : "memory", "scc");                                                                                      \
    }                                                                                                              \
  }

  f32x4 acc[4][NF];
  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  // this lane's four pixels of the tile at m_base: the class-(0,0) output pixel index (kOob-marked when the gy pixel does not exist)
  // and the edge bits, 2 per fragment f: 1 bottom row (dy = 1 leaves the image), 2 right column (dx = 1)
  uint32_t pix00[4];
  auto tile_pixels = [&](int m_base) {
    uint32_t eb = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint32_t m = (uint32_t)(m_base + wr * 64 + f * 16 + i16);
      const uint32_t img = fastdiv(m, d_howo), rem = m - img * d_howo.div;
      const uint32_t y = fastdiv(rem, d_wo), x = rem - y * d_wo.div;
      eb |= ((y == (uint32_t)Ho - 1 ? 1u : 0u) | (x == (uint32_t)Wo - 1 ? 2u : 0u)) << (2 * f);
      pix00[f] = (int)m < rows_total ? (img * (uint32_t)p.ho + 2u * y) * (uint32_t)Wi + 2u * x : 0xffffffffu;
    }
    return eb;
  };
  uint32_t edge8 = tile_pixels(m0);
  uint32_t ab[4];  // [2 dy + dx]: tile row wr*64 + i16 reads window row r + dy Wo + dx
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int joff = ((c & 2) ? Wo : 0) + (c & 1);
    const int row = wr * 64 + i16 + joff;
    ab[c] = (uint32_t)(row * 128 + ((g ^ ((i16 + joff) & 7)) << 4));
  }
  const uint32_t bb0 = (uint32_t)((wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128 + ((g ^ (i16 & 7)) << 4));
  // (K half 1 of a weight fragment: chunk (g + 4) ^ key = (g ^ key) ^ 4, i.e. bb0 with bit 6 flipped)

  u32x4 fa[2][2][4], fb[2][2][NF];  // [register set][K half][fragment]
#define VDQN_D_LOAD_FRAGS(SET, WBUF_, DY_, DX_, BBUF_)                                                             \
  {                                                                                                                \
    constexpr uint32_t tb_ = ((DY_) ? 1u : 0u) | ((DX_) ? 2u : 0u);                                                \
    const unsigned char* wb_ = smem + G::WinBase + (WBUF_)*kD_WinStride;                                           \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                             \
      uint32_t a0_ = ab[2 * (DY_) + (DX_)];                                                                        \
      if constexpr (tb_ != 0u) {                                                                                   \
        const bool z_ = (edge8 & (tb_ << (2 * f_))) != 0u;                                                         \
        a0_ = z_ ? ((a0_ & 255u) | (uint32_t)((kD_WinRows - 2) * 128 - f_ * 16 * 128)) : a0_;                      \
      }                                                                                                            \
      const uint32_t a1_ = a0_ ^ 64u;                                                                              \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a0_);                                 \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a1_);                                 \
    }                                                                                                              \
    const unsigned char* bt_ = smem + (BBUF_)*G::WtTile;                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                            \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + bb0);                                  \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + (bb0 ^ 64u));                          \
    }                                                                                                              \
  }
#define VDQN_D_MFMA_ALL(SET)                                                                                       \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)               \
      _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                          \
    acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[SET][h_][j_]),             \
                                                          __builtin_bit_cast(bf16x8, fa[SET][h_][f_]), acc[f_][j_], 0, 0, 0); \
  }
  // K-step U (0 .. 2 L - 1) of a body of phase KIND over the chunk pair (C0_, C0_ + 1): chunk C0_ + U / L, tap d_tap(KIND, U % L),
  // window buffer U / L, weight buffer / register set U & 1 (every body has an even step count).  It stages step U + 2 — inside the
  // body at compile time; the two steps behind its end from the run-time descriptors nx* (the next chunk pair of this phase, the
  // first steps of the next phase, or of the next tile) — and reads the fragments of step U + 1 under its own MFMAs.
#define VDQN_D_USTEP(KIND, U, C0_, NK_, NC_, NT_)                                                                               \
  {                                                                                                                \
    constexpr int L_ = d_len(KIND);                                                                                \
    constexpr int cur_ = (U)&1, nxt_ = cur_ ^ 1;                                                                   \
    constexpr int v_ = (U) + 2;              /* the step staged now */                                             \
    constexpr int w_ = ((U) + 1) % (2 * L_); /* the step whose fragments are read now (behind the body: its step 0 pattern) */ \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
    asm volatile("" : "+v"(edge8));                                                                                \
    asm volatile("" : "+v"(fa[cur_][0][0]), "+v"(fa[cur_][0][1]), "+v"(fa[cur_][0][2]), "+v"(fa[cur_][0][3]),      \
                      "+v"(fa[cur_][1][0]), "+v"(fa[cur_][1][1]), "+v"(fa[cur_][1][2]), "+v"(fa[cur_][1][3]));     \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[cur_][0][j_]), "+v"(fb[cur_][1][j_])); \
    __builtin_amdgcn_s_barrier();                                                                                  \
    if constexpr (v_ < 2 * L_) {                                                                                   \
      if constexpr ((KIND) == 4) {                                                                                 \
        VDQN_D_ISSUE_B(cur_, b2_off0, rs_b2, b2_n0 + ((C0_) + v_ / L_) * 128, b2_row32)                            \
      } else {                                                                                                     \
        VDQN_D_ISSUE_B(cur_, b_off0, rs_b, b_n0 + d_tap(KIND, v_ % L_).tap * tap_k + ((C0_) + v_ / L_) * 128, b_row32) \
      }                                                                                                            \
      if constexpr (v_ % L_ == 0) VDQN_D_ISSUE_AW(v_ / L_, m0, ((KIND) == 4 ? rs_a2 : rs_a), ((C0_) + v_ / L_) * 128) \
    } else { /* behind the body: step e_ (0 / 1) of a body of phase NK_ over the chunk pair (NC_, NC_ + 1), of this tile or (NT_) the next */ \
      constexpr int e_ = v_ - 2 * L_;                                                                              \
      constexpr int l2_ = d_len(NK_);                                                                              \
      const int nc_ = (NC_) + (l2_ == 1 ? e_ : 0);                                                                 \
      if constexpr ((NK_) == 4) {                                                                                  \
        VDQN_D_ISSUE_B(cur_, b2_off0, rs_b2, ((NT_) ? b2_n0_nx : b2_n0) + nc_ * 128, b2_row32)                     \
      } else {                                                                                                     \
        VDQN_D_ISSUE_B(cur_, b_off0, rs_b, ((NT_) ? b_n0_nx : b_n0) + d_tap(NK_, l2_ == 1 ? 0 : e_).tap * tap_k + nc_ * 128, b_row32) \
      }                                                                                                            \
      if constexpr (e_ == 0 || l2_ == 1) VDQN_D_ISSUE_AW(e_, ((NT_) ? m0_nx : m0), ((NK_) == 4 ? rs_a2 : rs_a), nc_ * 128) \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    VDQN_D_LOAD_FRAGS(nxt_, w_ / L_, d_tap(KIND, w_ % L_).dy, d_tap(KIND, w_ % L_).dx, nxt_)                       \
    VDQN_D_MFMA_ALL(cur_)                                                                                          \
    VDQN_INTERLEAVE(8 + 2 * NF)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }
#define VDQN_D_BODY(KIND, C0_, NK_, NC_, NT_)                                                                      \
  {                                                                                                                \
    VDQN_D_USTEP(KIND, 0, C0_, NK_, NC_, NT_) VDQN_D_USTEP(KIND, 1, C0_, NK_, NC_, NT_)                            \
    if constexpr (d_len(KIND) >= 2) { VDQN_D_USTEP(KIND, 2, C0_, NK_, NC_, NT_) VDQN_D_USTEP(KIND, 3, C0_, NK_, NC_, NT_) } \
    if constexpr (d_len(KIND) >= 4) {                                                                              \
      VDQN_D_USTEP(KIND, 4, C0_, NK_, NC_, NT_) VDQN_D_USTEP(KIND, 5, C0_, NK_, NC_, NT_)                          \
      VDQN_D_USTEP(KIND, 6, C0_, NK_, NC_, NT_) VDQN_D_USTEP(KIND, 7, C0_, NK_, NC_, NT_)                          \
    }                                                                                                              \
  }
  // a phase: its chunk pairs; the last pair (a copy of its own: what it stages behind its end is known at compile time) stages the
  // first steps of phase NK_ (of the next tile if NT_)
#define VDQN_D_PHASE(KIND, NK_, NT_)                                                                               \
  _Pragma("clang loop unroll(disable)") for (int it_ = 0; it_ + 1 < n_it; ++it_) VDQN_D_BODY(KIND, 2 * it_, KIND, 2 * it_ + 2, false) \
  VDQN_D_BODY(KIND, 2 * (n_it - 1), NK_, 0, NT_)

  int b_n0_nx = b_n0, b2_n0_nx = b2_n0, m0_nx = m0;  // the next tile's column-tile offsets and first pixel

  const LeanEpiD led = make_lean_epi_d(p.out, p.resid, p.mask, p.colsum_part, (long long)p.M, p.ldo, p.co);
  float* scratch = reinterpret_cast<float*>(smem + G::Scratch);
  auto epilogue = [&](int ph, int pw, int phase) {
    const int ncol = n0 + wc * (BN / WN) + g * CPL;
    const uint32_t shift = (uint32_t)(ph * Wi + pw);
    uint32_t off[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) off[f] = pix00[f] != 0xffffffffu ? ((pix00[f] + shift) * (uint32_t)p.ldo + (uint32_t)ncol) * 2u : kOob;
    lean_epilogue_dgrad<NF>(led, acc, scratch, off, n0, tile_m * 4 + phase, tid);
  };
#define VDQN_D_ZERO_ACC()                                                                    \
  _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) acc[f_][j_] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // end of an accumulation phase: what the last two steps staged has landed (for every wave), then the class's epilogue
#define VDQN_D_PHASE_END(PH_, PW_, PHASE_)                                                   \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                \
  __builtin_amdgcn_s_barrier();                                                              \
  epilogue(PH_, PW_, PHASE_);

  const int n_it = cpk >> 1;
  // prologue of a workgroup's first tile: steps 0 and 1 of a body of phase KIND over chunks (0, 1)
#define VDQN_D_PROLOGUE(KIND)                                                                \
  VDQN_D_ISSUE_B(0, b_off0, rs_b, b_n0 + d_tap(KIND, 0).tap * tap_k, b_row32)                \
  VDQN_D_ISSUE_AW(0, m0, rs_a, 0)                                                            \
  VDQN_D_ISSUE_B(1, b_off0, rs_b, b_n0 + d_tap(KIND, 1).tap * tap_k + (d_len(KIND) == 1 ? 128 : 0), b_row32) \
  if constexpr (d_len(KIND) == 1) VDQN_D_ISSUE_AW(1, m0, rs_a, 128)                          \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
  __builtin_amdgcn_s_barrier();
  // top of a tile: where the walk goes next (the last steps of this tile stage that tile's first two)
#define VDQN_D_TILE_TOP()                                                                    \
  const uint32_t lt_nx = lt + lt_step;                                                       \
  const bool has_nx = lt_nx < x_count;                                                       \
  const int tn_nx = has_nx ? (int)((x_first + lt_nx) % (uint32_t)tiles_n) : tile_n;          \
  const int tm_nx = has_nx ? (int)((x_first + lt_nx) / (uint32_t)tiles_n) : tile_m;          \
  m0_nx = tm_nx * BM;                                                                        \
  b_n0_nx = tn_nx * BN * 9 * tap_k;                                                          \
  b2_n0_nx = tn_nx * BN * tap_k;
#define VDQN_D_TILE_NEXT()                                                                   \
  if (!has_nx) break;                                                                        \
  lt = lt_nx;                                                                                \
  tile_n = tn_nx; tile_m = tm_nx;                                                            \
  n0 = tile_n * BN; m0 = m0_nx;                                                              \
  b_n0 = b_n0_nx;                                                                            \
  b2_n0 = b2_n0_nx;                                                                          \
  edge8 = tile_pixels(m0);
#define VDQN_D_CLASS_BEGIN(KIND)                                                             \
  VDQN_D_ZERO_ACC()                                                                          \
  VDQN_D_LOAD_FRAGS(0, 0, d_tap(KIND, 0).dy, d_tap(KIND, 0).dx, 0)

  if constexpr (!SPLIT) {
    VDQN_D_PROLOGUE(0)
    for (;;) {  // tiles of this workgroup: the four classes of each
      VDQN_D_TILE_TOP()
      VDQN_D_CLASS_BEGIN(0)  // class (1,1): four taps per chunk
      VDQN_D_PHASE(0, 1, false)
      VDQN_D_PHASE_END(1, 1, 0)
      VDQN_D_CLASS_BEGIN(1)  // class (1,0)
      VDQN_D_PHASE(1, 2, false)
      VDQN_D_PHASE_END(1, 0, 1)
      VDQN_D_CLASS_BEGIN(2)  // class (0,1)
      VDQN_D_PHASE(2, 3, false)
      VDQN_D_PHASE_END(0, 1, 2)
      VDQN_D_CLASS_BEGIN(3)  // class (0,0): the centre tap, then (SIB) the 1x1's chunks on the same accumulators
      if constexpr (SIB) {
        VDQN_D_PHASE(3, 4, false)
        VDQN_D_PHASE(4, 0, true)
      } else {
        VDQN_D_PHASE(3, 0, true)
      }
      VDQN_D_PHASE_END(0, 0, 3)
      VDQN_D_TILE_NEXT()
    }
  } else if (grp == 0) {
    VDQN_D_PROLOGUE(0)
    for (;;) {
      VDQN_D_TILE_TOP()
      VDQN_D_CLASS_BEGIN(0)
      VDQN_D_PHASE(0, 0, true)
      VDQN_D_PHASE_END(1, 1, 0)
      VDQN_D_TILE_NEXT()
    }
  } else if (grp == 1) {
    VDQN_D_PROLOGUE(1)
    for (;;) {
      VDQN_D_TILE_TOP()
      VDQN_D_CLASS_BEGIN(1)
      VDQN_D_PHASE(1, 2, false)
      VDQN_D_PHASE_END(1, 0, 1)
      VDQN_D_CLASS_BEGIN(2)
      VDQN_D_PHASE(2, 1, true)
      VDQN_D_PHASE_END(0, 1, 2)
      VDQN_D_TILE_NEXT()
    }
  } else {
    VDQN_D_PROLOGUE(3)
    for (;;) {
      VDQN_D_TILE_TOP()
      VDQN_D_CLASS_BEGIN(3)
      if constexpr (SIB) {
        VDQN_D_PHASE(3, 4, false)
        VDQN_D_PHASE(4, 3, true)
      } else {
        VDQN_D_PHASE(3, 3, true)
      }
      VDQN_D_PHASE_END(0, 0, 3)
      VDQN_D_TILE_NEXT()
    }
  }
#undef VDQN_D_CLASS_BEGIN
#undef VDQN_D_TILE_NEXT
#undef VDQN_D_TILE_TOP
#undef VDQN_D_PROLOGUE
#undef VDQN_D_PHASE_END
#undef VDQN_D_ZERO_ACC
#undef VDQN_D_PHASE
#undef VDQN_D_BODY
#undef VDQN_D_USTEP
#undef VDQN_D_MFMA_ALL
#undef VDQN_D_LOAD_FRAGS
#undef VDQN_D_ISSUE_B
#undef VDQN_D_ISSUE_AW
}

// the longest chain of K-steps a workgroup runs: every tile with all classes (one walk of xb workgroups over t tiles per XCD) ...
inline long d_chain_all(long t, long xb, int cpk, bool sib) { return ((t + xb - 1) / xb) * (long)cpk * (9 + (sib ? 1 : 0)); }
// ... and split into groups A : B : C = n_a : n_a : xb - 2 n_a workgroups (4 cpk, 4 cpk, (1 + sib) cpk steps per tile)
inline long d_chain_split(long t, long xb, long n_a, int cpk, bool sib) {
  const long n_c = xb - 2 * n_a;
  const long a = ((t + n_a - 1) / n_a) * 4 * cpk, c = ((t + n_c - 1) / n_c) * (long)cpk * (1 + (sib ? 1 : 0));
  return a > c ? a : c;
}

int g_s2d_split_override = -2;  // tests: vdqn_debug_set_s2d_split

template <int NF, bool SIB>
void launch_win9d(const IgemmParams& p, unsigned tiles, int tiles_n, hipStream_t stream) {
  using G = DGeom<NF>;
  static const int split_env0 = [] { const char* e = getenv("VDQN_S2DGRAD_SPLIT"); return e ? atoi(e) : -1; }();  // -1: by chain length
  const int split_env = g_s2d_split_override != -2 ? g_s2d_split_override : split_env0;
  const unsigned resident = 2u * (unsigned)vdqn_num_cus();
  const FastDiv d_wo = make_fastdiv((uint32_t)p.wi), d_howo = make_fastdiv((uint32_t)(p.hi * p.wi));
  const int cpk = p.ci / 64;
  // split launch: a grid of whole XCD rounds (every XCD the same number of workgroups), at most one workgroup per (tile, group)
  const unsigned want = 3u * tiles < resident ? 3u * tiles : resident;
  const long xb = want / 8, t_x = (tiles + 7) / 8;  // workgroups and (the larger) tile count per XCD
  long best_na = 0, best = 0;
  if (xb >= 3 && split_env != 0) {
    for (long n_a = 1; 2 * n_a < xb; ++n_a) {
      const long c = d_chain_split(t_x, xb, n_a, cpk, SIB);
      if (best_na == 0 || c < best) { best = c; best_na = n_a; }
    }
  }
  const unsigned grid_all = tiles > resident ? resident : tiles;
  const long all = d_chain_all(t_x, (grid_all + 7) / 8, cpk, SIB);
  // (-1: split where the chains get shorter AND a tile has enough K-steps to be bound by them — with two chunks per tile (layer2.0: 20 steps
  // beside four epilogues of 3 x 16 KB each) the launch is HBM-bound, and splitting the classes of a pixel block over workgroups that
  // run at different times cost 4 % there (profiles/r05o_bench_s2d_*.txt))
  if (best_na > 0 && (split_env == 1 || (best < all && cpk >= 4))) {
    vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9d_kernel<NF, SIB, true>), (size_t)G::Smem);
    hipLaunchKernelGGL((win9d_kernel<NF, SIB, true>), dim3((unsigned)(8 * xb)), dim3(256), G::Smem, stream, p, d_wo, d_howo, tiles, tiles_n, (uint32_t)best_na);
    return;
  }
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9d_kernel<NF, SIB, false>), (size_t)G::Smem);
  hipLaunchKernelGGL((win9d_kernel<NF, SIB, false>), dim3(grid_all), dim3(256), G::Smem, stream, p, d_wo, d_howo, tiles, tiles_n, 0u);
}

}  // namespace

// tests: force the class-group split of win9d_kernel on (1) / off (0), -1 = by chain length, -2 = back to VDQN_S2DGRAD_SPLIT
extern "C" void vdqn_debug_set_s2d_split(int v) { g_s2d_split_override = v; }

// whether vdqn_launch_win9d takes a stride-2 data-gradient call (igemm.hip asks; the geometry checks are the caller's): gy channels in
// chunk pairs, whole 64- or 128-column tiles, the lean epilogue's operand set
int vdqn_win9d_supports(int ci, int co, int has_sib, int ci2) { return ci % 128 == 0 && co % 64 == 0 && (!has_sib || ci2 == ci); }

// entry used by vdqn_conv2d (igemm.hip) for the data gradient of 3x3 / stride 2 / pad 1 over an even-sized image, bf16;
// p.in2 != nullptr: + the data gradient of the sibling 1x1 / stride-2 convolution (p.in2 / wt2)
int vdqn_launch_win9d(const void* pv, hipStream_t stream) {
  IgemmParams p = *reinterpret_cast<const IgemmParams*>(pv);
  const bool sib = p.in2 != nullptr;
  const int bn = p.co % 128 == 0 ? 128 : 64;
  const int tiles_n = p.co / bn;
  const int rows = p.n_img * p.hi * p.wi;
  const unsigned tiles = (unsigned)(((rows + 127) / 128) * tiles_n);
  p.wt_bytes = (int)((long long)p.co * p.ktot * 2);
  if (sib) p.wt2_bytes = (int)((long long)p.co * p.ci2 * 2);
  vdqn_prof_begin(bn == 128 ? "igemm_s2win<bf16,128,dgrad>" : "igemm_s2win<bf16,64,dgrad>", 2.0 * rows * p.co * p.ktot + (sib ? 2.0 * rows * p.co * p.ci2 : 0.0),
                  2.0 * ((double)rows * p.ci * (sib ? 2 : 1) + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))), stream);
  if (bn == 128) { if (sib) launch_win9d<4, true>(p, tiles, tiles_n, stream); else launch_win9d<4, false>(p, tiles, tiles_n, stream); }
  else { if (sib) launch_win9d<2, true>(p, tiles, tiles_n, stream); else launch_win9d<2, false>(p, tiles, tiles_n, stream); }
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
