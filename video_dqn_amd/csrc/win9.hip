// Nine-tap window kernel, unrolled: the 3x3 / stride 1 / pad 1 convolutions with 128+ channels (ResNet layer2 - layer4, bf16;
// MODE 0 forward, MODE 1 data gradient) — the dominant kernel of a TD update.
//
// Same tiling and staging as igemm_win9_kernel (igemm.hip): 128 x 128 tiles, 4 waves of 64 x 64, ONE staged window of
// 128 + 2 W + 2 consecutive input pixels per 64-channel chunk serving all nine taps, weight tiles streamed per K-step, two LDS
// buffers and two register sets of fragments, one barrier per K-step, two workgroups per CU.  What changed is the instruction
// stream of the K loop.  Counters and K-step stamps of the round-1 kernel (profiles/r02a_pmc_mfma.json,
// profiles/r02b_win9_kstep_stamps_clock.txt) showed a K-step issuing ~170 instructions per wave for its 32 MFMAs — ~85 scalar
// (tap decoding, the issue / load state machines, LDS-DMA address set-up with wait states), ~31 vector (swizzle keys, zero-row
// selects, address adds) — and a wave issues at most one instruction per 4 cycles: two waves per SIMD need ~1360 issue cycles
// for 1024 cycles of matrix work, which is the ~1500 cycles a K-step of the pair takes.  Here the K loop is unrolled over the
// 18 K-steps of a chunk pair, so every step knows its tap, its LDS buffers and its register set at compile time:
//   * per-lane fragment addresses of the nine taps (row offset W ky + kx and the XOR swizzle key of that row) are computed once
//     per workgroup (18 VGPRs); the window buffer, the fragment row f and the weight-ring slot are ds_read immediates;
//   * the zeros of an edge lane are selected with one v_and_or + one v_cndmask per fragment pair (a zero PAIR of rows, read at the
//     lane's own bank position: conflict-free); the centre tap and taps that cannot leave the image on a side skip the test;
//   * scalar work per step: the weight tile's K offset (one add), M0 for the LDS-DMA pieces.
// ~85 instructions per wave and K-step instead of ~170.
//
// Serves the same reference call sites as igemm.hip: torch conv2d (+ folded BatchNorm, ReLU, residual) of
// archs/HabitatDQNMultiAction.py:30,49-51 (torchvision BasicBlock conv1 / conv2) and their data gradient
// (train_q_network.py:226).
#include <stdlib.h>

#include "igemm_common.h"

namespace {

constexpr int kU_WtTile = 128 * 128;       // one staged weight tile
constexpr int kU_WinBase = 2 * kU_WtTile;  // LDS: [2 weight tiles][2 windows]

// BM = 128: 4 waves, two workgroups per CU.  BM = 256: 8 waves (4 x 2 of 64 x 64), one workgroup per CU — the two co-resident
// 128-row tiles of a CU made one, so that the weight tile (16 of the 18.7 KB a 128-row tile stages per K-step) is staged once for
// both halves: 21.4 KB per K-step and CU instead of 37.4 KB.
template <int BM>
struct Win9Geom {
  static constexpr int NT = 2 * BM;                     // threads
  static constexpr int RPP = NT / 8;                    // rows one staging pass of the workgroup covers (8 lanes x 16 B per row)
  static constexpr int PSTR = RPP * 128;                // LDS distance between a thread's consecutive DMA pieces
  static constexpr int WinRows = BM == 128 ? 192 : 320;  // >= BM + 2 * 28 + 3, a multiple of RPP
  static constexpr int WinStride = WinRows * 128;       // bytes between the two window buffers
  static constexpr int WPass = WinRows / RPP;           // 6 / 5 staging passes per window
  static constexpr int BPass = 128 / RPP;               // 4 / 2 per weight tile
  static constexpr int Smem = kU_WinBase + 2 * WinStride;
};

template <int MODE, int BM>
__global__ __launch_bounds__(2 * BM, 2) void win9u_kernel(const IgemmParams p, const int wrows, const FastDiv d_wo, const FastDiv d_howo, const uint32_t total_tiles, void* stamps) {
  static_assert(MODE == 0 || MODE == 1, "window kernel: forward or stride-1 data gradient");
  static_assert(BM == 128 || BM == 256, "tile rows");
  using T = bf16raw;  // (VDQN_INTERLEAVE keys on sizeof(T))
  using G = Win9Geom<BM>;
  constexpr int BN = 128, WN = 2;
  constexpr int NF = BN / (16 * WN);  // 4
  constexpr int CPL = 4 * NF;         // 16
  constexpr int PSTR = G::PSTR;
  constexpr int kU_WinStride = G::WinStride;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // Tiles of this workgroup.  The launch has either one workgroup per tile or (VDQN_WIN9_PERSIST) as many as the chip holds at
  // once, each walking several tiles: workgroups of XCD x = blockIdx & 7 own that XCD's contiguous range of logical tiles
  // (xcd_remap's ranges), workgroup j of the XCD takes tiles j, j + nb_x, j + 2 nb_x, ... of the range.
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t tq = total_tiles >> 3, tr = total_tiles & 7u;
  const uint32_t x_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;  // first logical tile of this XCD
  const uint32_t x_count = tq + (xcd < tr ? 1u : 0u);
  const uint32_t x_blocks = (gridDim.x >> 3) + (xcd < (gridDim.x & 7u) ? 1u : 0u);       // workgroups on this XCD
  uint32_t lt = blockIdx.x >> 3;                                                          // index inside the XCD's range
  int tile_n, tile_m, n0, m0;
  const int m_end = p.M;
  if (lt >= x_count) return;
  tile_n = (int)((x_first + lt) % (uint32_t)p.tiles_n);
  tile_m = (int)((x_first + lt) / (uint32_t)p.tiles_n);
  n0 = tile_n * BN;
  m0 = tile_m * BM;
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

  // ---- window rows staged by this thread: j = lrow + RPP i; rows past BM + 2 W + 2 (and pixels outside the tensor) are zero.
  // The offsets are rebuilt from (q0, lchunk) at every window issue (once per nine K-steps) instead of held in registers:
  // the K loop needs the VGPRs for two fragment sets and the per-tap addresses ----
  const int pixB = p.pix_stride * 2;
  const int need = BM + 2 * W + 2;
  int q0 = m0 - W - 1 + lrow;  // input pixel of window row lrow (of the current tile)
  const uint32_t a_lane = (uint32_t)(lchunk_a * 16);
  // weight rows lrow + RPP i: one per-lane offset, the row stride goes into the DMA's scalar offset (the weight tensor holds all
  // 128 rows of the column tile, so no range check is involved)
  uint32_t b_off0 = (uint32_t)(n0 + lrow) * (uint32_t)(p.ktot * 2) + (uint32_t)(lchunk_b * 16);
  const int b_row32 = G::RPP * p.ktot * 2;

  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * (8 * 128);

  // LDS-DMA from inline asm (hipcc would wait vmcnt(0) before the first ds_read behind a pending LDS-DMA); M0 = LDS address of
  // the wave's piece, one wait state between the M0 write and the DMA that reads it
#define VDQN_DMA4(V0, V1, V2, V3, LDS, RSRC, SOFF)                                                                  \
  asm volatile(                                                                                                     \
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %5, %6 offen lds\n\t"                                \
      "s_add_u32 m0, %4, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"                            \
      "s_add_u32 m0, %4, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds"                                 \
      ::"v"(V0), "v"(V1), "v"(V2), "v"(V3), "s"(LDS), "s"(RSRC), "s"(SOFF), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
      : "memory", "scc")
  // activation window of channel chunk CC -> window buffer WBUF: WPass passes of RPP rows (a window buffer has WinRows rows whatever
  // W is; rows past BM + 2 W + 2 get an out-of-range offset and are zero-filled, the last of them is the zero row)
#define VDQN_ISSUE_AW(WBUF, SO_A, Q0)                                                                                   \
  {                                                                                                                 \
    const uint32_t la_ = lds_wave + (uint32_t)(kU_WinBase + (WBUF)*kU_WinStride);                                   \
    const int so_a_ = (SO_A);                                                                                       \
    int q_ = (Q0);                                                                                                  \
    asm volatile("" : "+v"(q_)); /* rebuilt here, not hoisted into loop-carried registers */                         \
    uint32_t a_off[G::WPass];                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < G::WPass; ++i_) {                                                       \
      const int qi_ = q_ + G::RPP * i_;                                                                             \
      a_off[i_] = (lrow + G::RPP * i_ < need && (unsigned)qi_ < (unsigned)rows_total) ? (uint32_t)qi_ * (uint32_t)pixB + a_lane : kOob; \
    }                                                                                                               \
    VDQN_DMA4(a_off[0], a_off[1], a_off[2], a_off[3], la_, rs_a, so_a_);                                            \
    const uint32_t l4_ = la_ + 4 * PSTR;                                                                            \
    if constexpr (G::WPass == 6) {                                                                                  \
      asm volatile(                                                                                                 \
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %4 offen lds\n\t"                            \
          "s_add_u32 m0, %2, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds"                             \
          ::"v"(a_off[4]), "v"(a_off[G::WPass - 1]), "s"(l4_), "s"(rs_a), "s"(so_a_), "n"(PSTR)                     \
          : "memory", "scc");                                                                                       \
    } else {                                                                                                        \
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"                       \
                   ::"v"(a_off[4]), "s"(l4_), "s"(rs_a), "s"(so_a_) : "memory");                                    \
    }                                                                                                               \
  }
#define VDQN_ISSUE_B(BUF, SOFF, BOFF, RSB)                                                                              \
  {                                                                                                                 \
    const i32x4 rs_sel_ = (RSB);                                                                                    \
    const uint32_t lb_ = lds_wave + (uint32_t)((BUF)*kU_WtTile);                                                    \
    const int so0_ = (SOFF), so1_ = so0_ + b_row32;                                                                 \
    if constexpr (G::BPass == 4) {                                                                                  \
      const int so2_ = so1_ + b_row32, so3_ = so2_ + b_row32;                                                       \
      asm volatile(                                                                                                 \
          "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\t"                            \
          "s_add_u32 m0, %1, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %4 offen lds\n\t"                        \
          "s_add_u32 m0, %1, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %5 offen lds\n\t"                        \
          "s_add_u32 m0, %1, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %6 offen lds"                             \
          ::"v"(BOFF), "s"(lb_), "s"(rs_sel_), "s"(so0_), "s"(so1_), "s"(so2_), "s"(so3_), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
          : "memory", "scc");                                                                                       \
    } else {                                                                                                        \
      asm volatile(                                                                                                 \
          "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\t"                            \
          "s_add_u32 m0, %1, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %4 offen lds"                             \
          ::"v"(BOFF), "s"(lb_), "s"(rs_sel_), "s"(so0_), "s"(so1_), "n"(PSTR)                                      \
          : "memory", "scc");                                                                                       \
    }                                                                                                               \
  }

  // forward tiles with bias (+ residual, ReLU) and nothing else take the lean epilogue (igemm_common.h); f32
  // copies, ragged rows and every data-gradient operand combination keep igemm_epilogue
  const bool lean = MODE == 0 && !p.no_lean && p.co % BN == 0 && p.vec_ok && p.out && !p.out_f32 && !p.colsum_part && !p.mask && p.bias &&
                    (((uintptr_t)p.bias) & 15) == 0 && (long long)p.M * p.ldo * 2 < 0x7fffffffLL;
  const bool lean_d = MODE == 1 && !p.no_lean && p.vec_ok && p.out && !p.out_f32 && !p.bias && p.co % BN == 0 && (long long)p.M * p.ldo * 2 < 0x7fffffffLL;
  const LeanEpiD led = make_lean_epi_d(p.out, p.resid, p.mask, p.colsum_part, lean_d ? p.M : 0, p.ldo, p.co);
  const LeanEpi le = make_lean_epi(p.out, p.resid, p.bias, lean ? p.M : 0, p.ldo, p.co, p.relu);
  f32x4 acc[4][NF];
  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  // edge bits of this lane's four pixels (of the tile at m_base), 4 bits per fragment f: 1 top row, 2 bottom row, 4 left
  // column, 8 right column
  auto edge_bits = [&](int m_base) {
    uint32_t eb = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint32_t m = (uint32_t)(m_base + wr * 64 + f * 16 + i16);
      const uint32_t rem = m - fastdiv(m, d_howo) * d_howo.div;
      const uint32_t oh = fastdiv(rem, d_wo), ow = rem - oh * d_wo.div;
      const uint32_t e = (oh == 0 ? 1u : 0u) | (oh == (uint32_t)H - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == (uint32_t)W - 1 ? 8u : 0u);
      eb |= e << (4 * f);
    }
    return eb;
  };
  uint32_t edge16 = edge_bits(m0);

  // ---- per-lane LDS byte offsets, constant over the K loop ----
  // ab[tap][h]: fragment row f = 0 of tap (kr, ks), K half h, relative to a window buffer: tile row wr*64 + i16 reads window row
  // r + W ky + kx (forward: (ky, kx) = (kr, ks); data gradient: (2 - kr, 2 - ks)); the 16-byte chunk g + 4 h sits at the position
  // XOR-ed with that window row's key (row & 7)
  // (K half 1 is the same address with bit 6 flipped: chunk (g + 4) ^ key = (g ^ key) ^ 4)
  uint32_t ab[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int kr = t / 3, ks = t % 3;
    const int ky = MODE == 0 ? kr : 2 - kr, kx = MODE == 0 ? ks : 2 - ks;
    const int row = wr * 64 + i16 + W * ky + kx;
    const int key = (i16 + W * ky + kx) & 7;  // (wr * 64 is a multiple of 8)
    ab[t] = (uint32_t)(row * 128 + ((g ^ key) << 4));
  }
  // zs[f]: the zero PAIR (the last two rows of a window buffer: 256 bytes at a 256-byte boundary = every LDS bank once), minus
  // the f * 16 rows the read's immediate adds.  An edge lane reads the zeros at ITS OWN position inside the pair (its address
  // modulo 256): the lanes of a ds_read_b128 group then still hit 16 different bank quads.  With one shared zero-row address per
  // K-chunk (round 2) every edge lane collided with a neighbour in each of its four lane groups — one extra LDS cycle per group,
  // i.e. a fragment read with an edge pixel took 8 cycles instead of 4: 27-43 % of the activation-fragment LDS cycles by the bank
  // model (left / right image borders fall into almost every 16-pixel fragment), 23 % of all LDS cycles by SQ_LDS_BANK_CONFLICT.
  uint32_t zs[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) zs[f] = (uint32_t)((wrows - 2) * 128 - f * 16 * 128);
  // bb[h]: weight fragment j = 0, relative to a weight tile (rows read in the permuted order the epilogue expects)
  const uint32_t bb0 = (uint32_t)((wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128 + ((g ^ (i16 & 7)) << 4));
  const uint32_t bb1 = (uint32_t)((wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128 + (((g + 4) ^ (i16 & 7)) << 4));

  const int cpk = p.ci / 64;   // channel chunks (even: ci is a multiple of 128); K order (chunk, tap), tap fastest
  const int n_it = cpk >> 1;   // iterations of the 18-step body
  const int tap_k = cpk * 128;  // byte distance between the weight K offsets of consecutive taps of one chunk

  u32x4 fa[2][2][4], fb[2][2][NF];  // [register set][K half][fragment]

  // fragments of the K-step with tap TAP_ in window buffer WBUF_ / weight buffer BBUF_ -> register set SET
#define VDQN_LOAD_FRAGS_N(SET, TAP_, WBUF_, BBUF_, FCNT_)                                                                \
  {                                                                                                                      \
    constexpr int kr_ = (TAP_) / 3, ks_ = (TAP_) % 3;                                                                    \
    constexpr int ky_ = MODE == 0 ? kr_ : 2 - kr_, kx_ = MODE == 0 ? ks_ : 2 - ks_;                                      \
    constexpr uint32_t tb_ = (ky_ == 0 ? 1u : 0u) | (ky_ == 2 ? 2u : 0u) | (kx_ == 0 ? 4u : 0u) | (kx_ == 2 ? 8u : 0u);   \
    const unsigned char* wb_ = smem + kU_WinBase + (WBUF_)*kU_WinStride;                                                 \
    _Pragma("unroll") for (int f_ = 0; f_ < (FCNT_); ++f_) {                                                             \
      uint32_t a0_ = ab[TAP_];                                                                                           \
      if constexpr (tb_ != 0u) { /* an edge lane's tap leaves the image: read the zero row */                            \
        const bool z_ = (edge16 & (tb_ << (4 * f_))) != 0u;                                                              \
        a0_ = z_ ? ((a0_ & 255u) | zs[f_]) : a0_;                                                                        \
      }                                                                                                                  \
      const uint32_t a1_ = a0_ ^ 64u;                                                                                    \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a0_);                                       \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a1_);                                       \
    }                                                                                                                    \
    const unsigned char* bt_ = smem + (BBUF_)*kU_WtTile;                                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                  \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + bb0);                                        \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + bb1);                                        \
    }                                                                                                                    \
  }
#define VDQN_LOAD_FRAGS(SET, TAP_, WBUF_, BBUF_) VDQN_LOAD_FRAGS_N(SET, TAP_, WBUF_, BBUF_, 4)
#define VDQN_MFMA_ALL(SET) VDQN_MFMA_N(SET, 4)
#define VDQN_MFMA_N(SET, FCNT_)                                                                                          \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < (FCNT_); ++f_)               \
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
  // K-step U (0..17) of the iteration over chunks 2 it, 2 it + 1: tap U % 9 of chunk 2 it + U / 9.  Its fragments are in register
  // set U & 1 (read one step ago); it issues the staging of step U + 2 (weight buffer U & 1, just released; at tap 0 also that
  // chunk's window) and reads the fragments of step U + 1 underneath its own MFMAs.
#define VDQN_USTEP(U)                                                                                                    \
  {                                                                                                                      \
    constexpr int cur_ = (U)&1, nxt_ = cur_ ^ 1;                                                                         \
    constexpr int ti_ = ((U) + 2) % 9, ci_ = ((U) + 2) / 9; /* tap and chunk (relative to 2 it) of the step staged now */  \
    constexpr int tl_ = ((U) + 1) % 9, cl_ = ((U) + 1) / 9; /* ... of the step whose fragments are read now */            \
    VDQN_ST(st_comp)                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                          \
    asm volatile("" : "+v"(edge16)); /* keeps the (loop-invariant) zero-row selects of this step inside this step */       \
    VDQN_ST(st_wait)                                                                                                     \
    asm volatile("" : "+v"(fa[cur_][0][0]), "+v"(fa[cur_][0][1]), "+v"(fa[cur_][0][2]), "+v"(fa[cur_][0][3]),            \
                      "+v"(fa[cur_][1][0]), "+v"(fa[cur_][1][1]), "+v"(fa[cur_][1][2]), "+v"(fa[cur_][1][3]));           \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[cur_][0][j_]), "+v"(fb[cur_][1][j_]));   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    VDQN_ST(st_bar)                                                                                                      \
    /* no branch: behind the last step this stages two tiles nobody reads (out-of-range reads are zero-filled) */      \
    /* steps 16 and 17 stage the first two K-steps of what comes next: chunk 2 it + 2 of this tile, or — in the tile's last     \
       iteration — chunk 0 of the workgroup's NEXT tile (so_nx / b_nx / q_nx), whose prologue thereby runs under this tile's   \
       last steps and epilogue */                                                                                         \
    if constexpr (ci_ == 2) {                                                                                            \
      VDQN_ISSUE_B(cur_, ti_ * tap_k + so_nx, b_nx, rs_b)                                                               \
      if constexpr (ti_ == 0) VDQN_ISSUE_AW(0, so_nx, q_nx)                                                              \
    } else {                                                                                                             \
      VDQN_ISSUE_B(cur_, ti_ * tap_k + (cc2 + ci_) * 128, b_off0, rs_b)                                                  \
      if constexpr (ti_ == 0) VDQN_ISSUE_AW(ci_ & 1, (cc2 + ci_) * 128, q0)                                              \
    }                                                                                                                    \
    VDQN_ST(st_issue)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    VDQN_LOAD_FRAGS(nxt_, tl_, cl_ & 1, nxt_) /* unconditional: the step behind the last one re-reads buffers that still exist */ \
    VDQN_MFMA_ALL(cur_)                                                                                                  \
    VDQN_INTERLEAVE(8 + 2 * NF)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }


  // prologue of the workgroup's FIRST tile: K-steps 0 and 1 (window of chunk 0, weight tiles of taps 0 and 1)
  VDQN_ISSUE_B(0, 0, b_off0, rs_b)
  VDQN_ISSUE_AW(0, 0, q0)
  VDQN_ISSUE_B(1, tap_k, b_off0, rs_b)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef VDQN_STAMP
  st_t = __builtin_amdgcn_s_memtime();
  unsigned long long st_loop_end = 0;
#endif
  for (;;) {  // tiles of this workgroup
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    VDQN_LOAD_FRAGS(0, 0, 0, 0)  // the fragments of step 0
    // the next tile of this workgroup (if any): where its first window and weight tiles come from
    const uint32_t lt_nx = lt + x_blocks;
    const bool has_nx = lt_nx < x_count;
    const int tn_nx = !has_nx ? tile_n : (int)((x_first + lt_nx) % (uint32_t)p.tiles_n);
    const int tm_nx = !has_nx ? tile_m : (int)((x_first + lt_nx) / (uint32_t)p.tiles_n);
    const int m0_nx = tm_nx * BM;
    const int q0_t = m0_nx - W - 1 + lrow;
    const uint32_t b_t = (uint32_t)(tn_nx * BN + lrow) * (uint32_t)(p.ktot * 2) + (uint32_t)(lchunk_b * 16);
#pragma clang loop unroll(disable)
    for (int it = 0; it < n_it; ++it) {
      const int cc2 = 2 * it;  // first chunk of this iteration
      // (without a next tile the steps behind the end stage this tile's chunk 2 it + 2: nobody reads it)
      const bool last_it = has_nx && it == n_it - 1;
      const int so_nx = last_it ? 0 : (cc2 + 2) * 128;
      const int q_nx = last_it ? q0_t : q0;
      const uint32_t b_nx = last_it ? b_t : b_off0;
      VDQN_USTEP(0) VDQN_USTEP(1) VDQN_USTEP(2) VDQN_USTEP(3) VDQN_USTEP(4) VDQN_USTEP(5)
      VDQN_USTEP(6) VDQN_USTEP(7) VDQN_USTEP(8) VDQN_USTEP(9) VDQN_USTEP(10) VDQN_USTEP(11)
      VDQN_USTEP(12) VDQN_USTEP(13) VDQN_USTEP(14) VDQN_USTEP(15) VDQN_USTEP(16) VDQN_USTEP(17)
    }
    VDQN_ST(st_comp)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two tiles staged behind the last step have landed (the next tile's
    __builtin_amdgcn_s_barrier();                     // K-steps 0 and 1: weight buffers 0, 1 and window buffer 0 stay untouched)
#ifdef VDQN_STAMP
    st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    // the epilogue's scratch (column sums) is window buffer 1: its last reader was step 16.  (m_end: rows behind M are not stored)
    if (MODE == 0 && lean) lean_epilogue_128<WN>(le, acc, m0, n0, m_end, tid);
    else if (MODE == 1 && lean_d)
      lean_epilogue_dgrad_128<WN, BM / 64>(led, acc, reinterpret_cast<float*>(smem + kU_WinBase + kU_WinStride), m0, n0, tile_m, m_end, tid);
    else igemm_epilogue<T, BM, BN, MODE, WN>(p, acc, smem + kU_WinBase + kU_WinStride, m0, n0, tile_m, m_end, p.howo, W, 0, 0,
                                             p.bias);
    if (!has_nx) break;
    lt = lt_nx;
    tile_n = tn_nx; tile_m = tm_nx;
    n0 = tile_n * BN; m0 = m0_nx;
    q0 = q0_t;
    b_off0 = b_t;
    edge16 = edge_bits(m0);
  }
#undef VDQN_USTEP
#undef VDQN_LOAD_FRAGS
#undef VDQN_MFMA_ALL
#undef VDQN_MFMA_N
#undef VDQN_LOAD_FRAGS_N
#undef VDQN_ISSUE_AW
#undef VDQN_ISSUE_B
#undef VDQN_DMA4
#ifdef VDQN_STAMP
  if (stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the epilogue's stores have left
    unsigned long long* o = reinterpret_cast<unsigned long long*>(stamps) + (size_t)blockIdx.x * 16;
    o[0] = st_begin; o[1] = st_loop_end; o[2] = __builtin_amdgcn_s_memtime();
    o[3] = st_wait; o[4] = st_bar; o[5] = st_issue; o[6] = st_comp; o[7] = (unsigned long long)p.nk;
    o[8] = st_rt_begin; o[9] = __builtin_amdgcn_s_memrealtime();  // 100 MHz reference clock
  }
#endif
#undef VDQN_ST
}

}  // namespace

#ifdef VDQN_STAMP
extern void* g_stamp_buffer;
#endif

static int g_win9_bm256_override = -1;
extern "C" void vdqn_debug_set_win9_bm256(int v) { g_win9_bm256_override = v; }  // test hook: -1 = VDQN_WIN9_BM256
// entry used by vdqn_conv2d (igemm.hip): returns VDQN_OK or an error code
template <int MODE, int BM>
static void launch_win9u(const IgemmParams& p, hipStream_t stream, void* stamps) {
  using G = Win9Geom<BM>;
  const int wrows = (BM + 2 * p.wo + 2 + 1 + 7) & ~7;  // <= G::WinRows for W <= 28
  const unsigned tiles = (unsigned)(((p.M + BM - 1) / BM) * p.tiles_n);
  // VDQN_WIN9_PERSIST=1: at most as many workgroups as the chip holds at once, each walking its tiles with the next tile's first
  // two K-steps staged under the current tile's last steps and epilogue; 0: one workgroup per tile
  // (default: for launches of more than two rounds of resident workgroups — layer2 +5-7 %, layer3 at 512 frames +4 %; a launch of
  // 1.5 rounds loses 2-3 % to the static tile assignment; 2: always; profiles/r02o_win9u_persistent.txt)
  // The remainder policies built against the partial last round of this static walk — a balanced row walk, a split-K remainder
  // launch, whole rounds on 256-row tiles + the rest on 128-row tiles — all measured slower and live in experiments/ (DESIGN.md 6d).
  static const int persist = [] { const char* e = getenv("VDQN_WIN9_PERSIST"); return e ? atoi(e) : 1; }();
  const unsigned resident = (unsigned)((BM == 128 ? 2 : 1) * vdqn_num_cus());
  // (256-row tiles, one workgroup per CU: persistent above one round — VDQN_WIN9_PERSIST256=0: one workgroup per tile, as before round 5)
  static const int persist256 = [] { const char* e = getenv("VDQN_WIN9_PERSIST256"); return e ? atoi(e) : 1; }();
  const unsigned grid = ((BM == 128 || persist256) && ((persist == 1 && tiles > 2 * resident) || (persist >= 2 && tiles > resident) || (BM == 256 && persist256 && tiles > resident))) ? resident : tiles;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9u_kernel<MODE, BM>), (size_t)G::Smem);
  hipLaunchKernelGGL((win9u_kernel<MODE, BM>), dim3(grid), dim3(G::NT), G::Smem, stream, p, wrows, make_fastdiv((uint32_t)p.wo),
                     make_fastdiv((uint32_t)p.howo), tiles, stamps);
}

int vdqn_launch_win9u(const void* pv, int mode, hipStream_t stream) {
  const IgemmParams& p = *reinterpret_cast<const IgemmParams*>(pv);
  // VDQN_WIN9_BM256: 256-row tiles (eight waves, ONE workgroup per CU: a K-step stages 16 KB of weights for 256 rows instead of
  // for 128 — 21 instead of 37 KB per CU and step).  Per K-step they are the faster form from four channel chunks on, but a launch
  // of r rounds costs them ceil(r) full rounds, while the 128-row tiles' last partial round is cheap (its workgroups run alone on
  // their CUs): measured per layer and batch (profiles/r05z_bench_conv_*.txt) they win where the fractional part of
  // r = tiles / resident is >= ~0.5 (layer4 at 256 / 512 frames: +4 / +8 %, layer3 at 256 / 768: +6 / +7 %) and lose where it is small
  // (layer3 at 512 frames, 3.06 rounds: -9 %; layer4 at 768, 2.30: -9 %) and on layer2 (two chunks) everywhere.
  // 3 (default) = by that rule, 2 = always, 1 = wherever the launch has more 128-row tiles than the chip holds at once, 0 = never
  static const int bm256_env = [] { const char* e = getenv("VDQN_WIN9_BM256"); return e ? atoi(e) : 3; }();
  const int bm256 = g_win9_bm256_override >= 0 ? g_win9_bm256_override : bm256_env;
  bool big = bm256 == 2 || (bm256 == 1 && (long long)p.tiles_m * p.tiles_n > 2ll * vdqn_num_cus());
  // (the data gradient's 256-row tiles got the lean epilogue with this rule: on the shared igemm_epilogue they measured 0.617 vs
  // 0.612 ms in the update although the bare kernel is 3-5 % faster, profiles/r06a_ab_bm256_auto_vs_128.txt)
  if (bm256 == 3 && p.ci >= 256) {
    const long long tiles128 = (long long)p.tiles_m * p.tiles_n, res128 = 2ll * vdqn_num_cus();
    const double r = (double)tiles128 / (double)res128;  // rounds of the 128-row tiles (p.tiles_m counts those)
    const double frac = r - (double)(long long)r;
    // (layer4 at 192 frames — config 5's target and backward passes, 0.58 rounds — gains 4 % on 256-row tiles, every other layer at
    // 192 / 384 frames stays on 128 rows: profiles/r6_06_bench_conv_192_384_frames_tile_rule.txt)
    big = r >= (p.ci >= 512 ? 0.55 : 0.7) && frac >= 0.45;
  }
  void* stamps = nullptr;
#ifdef VDQN_STAMP
  stamps = g_stamp_buffer;
#endif
  vdqn_prof_begin(mode == 0 ? "igemm_win<bf16,128,fwd>" : "igemm_win<bf16,128,dgrad>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))), stream);
  if (mode == 0) {
    if (big) launch_win9u<0, 256>(p, stream, stamps); else launch_win9u<0, 128>(p, stream, stamps);
  } else {
    if (big) launch_win9u<1, 256>(p, stream, stamps); else launch_win9u<1, 128>(p, stream, stamps);
  }
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
