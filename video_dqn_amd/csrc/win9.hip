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

template <int MODE, int BM, int WALK = 0>
__global__ __launch_bounds__(2 * BM, 2) void win9u_kernel(const IgemmParams p, const int wrows, const FastDiv d_wo, const FastDiv d_howo, const uint32_t total_tiles, void* stamps,
                                                          const int bal_rows) {
  static_assert(MODE == 0 || MODE == 1, "window kernel: forward or stride-1 data gradient");
  static_assert(BM == 128 || BM == 256, "tile rows");
  constexpr bool BAL = WALK == 1 || WALK == 2;  // 1: the full tiles of every workgroup's row range, 2 (a second launch): their partial last tiles
  constexpr bool SK = WALK == 3;                // the launch's REMAINDER tiles (behind its whole rounds), split along K: see the SK section below
  static_assert(!BAL || (MODE == 0 && BM == 128), "balanced walk: forward, 128-row tiles");
  static_assert(!SK || BM == 128, "split-K remainder: 128-row tiles");
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
  //
  // BALANCED walk (WALK 1 + 2; round 5, forward without column sums): a launch of 3.06 rounds of resident workgroups costs four
  // tile times with the static walk above (0.77 of its work per round: tools/bench_win9.py).  Here every
  // workgroup owns ONE column tile and bal_rows consecutive output rows of it (M split evenly over the gridDim.x / tiles_n
  // workgroups of a column, rounded up to 16).  WALK 1 walks the range's full 128-row tiles.  The rest of the range — fewer than 128
  // rows — is the PARTIAL tile of WALK 2, a second launch of the same grid: a wave computes only the 16-row fragments that hold rows
  // of the range (all four, one, or none: three copies of the K loop, chosen per wave) through the same barriers and DMA pieces, and
  // the epilogue stores nothing behind the range's end.  (One kernel with both parts was built first: the extra loop copies pushed
  // the allocator into spilling a few per-lane addresses inside the main K loop, and a scratch reload there returns only behind
  // the LDS-DMA pieces issued before it — vmcnt retires in order — which cost 60 %: profiles/r05c_bench_win9_balanced_one_kernel.txt.)
  // Workgroups of one XCD take the column tiles of the same row ranges (the activation rows stay in that L2).
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t tq = total_tiles >> 3, tr = total_tiles & 7u;
  const uint32_t x_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;  // first logical tile of this XCD
  const uint32_t x_count = tq + (xcd < tr ? 1u : 0u);
  const uint32_t x_blocks = (gridDim.x >> 3) + (xcd < (gridDim.x & 7u) ? 1u : 0u);       // workgroups on this XCD
  uint32_t lt = blockIdx.x >> 3;                                                          // index inside the XCD's range
  int tile_n, tile_m, n0, m0;
  int m_end = p.M;  // rows behind it are not this workgroup's (balanced walk: the end of its row range)
  if constexpr (BAL) {
    const uint32_t per_xcd = gridDim.x >> 3;                      // (the launcher makes the grid a multiple of 8 tiles_n)
    const uint32_t j = blockIdx.x >> 3;
    tile_n = (int)(j % (uint32_t)p.tiles_n);
    const uint32_t k = xcd * (per_xcd / (uint32_t)p.tiles_n) + j / (uint32_t)p.tiles_n;  // row range of this workgroup
    const long long mb = (long long)k * bal_rows;
    if (mb >= p.M) return;
    m0 = (int)mb;
    m_end = (int)(mb + bal_rows < p.M ? mb + bal_rows : p.M);
    const int full = (m_end - m0) / BM * BM;  // rows of the range in full tiles
    if constexpr (WALK == 1) {
      m_end = m0 + full;
      if (full == 0) return;
    } else {
      m0 += full;
      if (m0 >= m_end) return;
    }
    tile_m = m0 / BM;  // (column sums are kept off this walk; the grouped forward too)
    n0 = tile_n * BN;
  } else if constexpr (SK) {
    // (remainder launch: total_tiles = the remainder's tile count, bal_rows = the index of its first tile; the items of this
    // workgroup are worked out in the SK section — start from the XCD's first remainder tile so that everything below is defined)
    if (x_count == 0) return;
    tile_n = (int)(((uint32_t)bal_rows + x_first) % (uint32_t)p.tiles_n);
    tile_m = (int)(((uint32_t)bal_rows + x_first) / (uint32_t)p.tiles_n);
    n0 = tile_n * BN;
    m0 = tile_m * BM;
  } else {
    if (lt >= x_count) return;
    // (a launch of whole rounds in front of a split-K remainder launch clears that launch's arrival counters)
    if (p.sk_cnt && blockIdx.x == 0 && tid < 256) reinterpret_cast<uint4*>(p.sk_cnt)[tid] = make_uint4(0u, 0u, 0u, 0u);
    // (static walk: bal_rows = index of the launch's first tile — a launch may cover a tail of the tile sequence, see the launcher)
    tile_n = (int)(((uint32_t)bal_rows + x_first + lt) % (uint32_t)p.tiles_n);
    tile_m = (int)(((uint32_t)bal_rows + x_first + lt) / (uint32_t)p.tiles_n);
    n0 = tile_n * BN;
    m0 = tile_m * BM;
  }
  const int W = p.wo, H = p.ho, rows_total = p.M;
  const int lrow = tid >> 3;
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int lchunk_b = (tid & 7) ^ ((((lrow / CPL) & 1) << 2) | (lrow & 3));

  const unsigned long long a_ptr = (unsigned long long)p.in;
  const unsigned long long b_ptr = (unsigned long long)p.wt;
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const i32x4 rs_b0 = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};
  // grouped forward (IgemmParams::m_split): tiles from row m_split on take the second weight set.  A persistent workgroup may walk
  // tiles of both sets, so the weight descriptor is chosen per tile (four scalar selects) — for the tile being computed (rs_b) and
  // for the next tile, whose first two weight tiles are staged under this tile's last steps (rs_bn).
  const unsigned long long bb_ptr = (unsigned long long)(MODE == 0 && p.wt_b ? p.wt_b : p.wt);
  const i32x4 rs_b1 = {__builtin_amdgcn_readfirstlane((int)(unsigned)bb_ptr), __builtin_amdgcn_readfirstlane((int)((bb_ptr >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};
  const int m_split = (MODE == 0 && !BAL && !SK) ? p.m_split : 0x7fffffff;
  i32x4 rs_b = m0 >= m_split ? rs_b1 : rs_b0;

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

  // forward tiles with bias (+ residual, ReLU) and nothing else take the lean epilogue (igemm_common.h); the grouped forward, f32
  // copies, ragged rows and every data-gradient operand combination keep igemm_epilogue
  const bool lean = MODE == 0 && !p.no_lean && p.co % BN == 0 && p.vec_ok && p.out && !p.out_f32 && !p.colsum_part && !p.mask && p.bias && !p.wt_b &&
                    (((uintptr_t)p.bias) & 15) == 0 && (long long)p.M * p.ldo * 2 < 0x7fffffffLL;
  const bool lean_d = MODE == 1 && !p.no_lean && p.vec_ok && p.out && !p.out_f32 && !p.bias && p.co % BN == 0 && (long long)p.M * p.ldo * 2 < 0x7fffffffLL;
  const LeanEpiD led = make_lean_epi_d(p.out, p.resid, p.mask, p.colsum_part, lean_d ? p.M : 0, p.ldo, p.co);
  const LeanEpi le = make_lean_epi(p.out, p.resid, p.bias, (BAL || lean) ? p.M : 0, p.ldo, p.co, p.relu);
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
        if constexpr (SK) asm volatile("" : "+v"(zs[f_])); /* (the remainder kernel has no room for 36 hoisted zero-row addresses) */ \
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
#define VDQN_USTEP(U, FCNT_)                                                                                             \
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
      VDQN_ISSUE_B(cur_, ti_ * tap_k + so_nx, b_nx, rs_bn)                                                               \
      if constexpr (ti_ == 0) VDQN_ISSUE_AW(0, so_nx, q_nx)                                                              \
    } else {                                                                                                             \
      VDQN_ISSUE_B(cur_, ti_ * tap_k + (cc2 + ci_) * 128, b_off0, rs_b)                                                  \
      if constexpr (ti_ == 0) VDQN_ISSUE_AW(ci_ & 1, (cc2 + ci_) * 128, q0)                                              \
    }                                                                                                                    \
    VDQN_ST(st_issue)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    if constexpr ((FCNT_) == 4) {                                                                                        \
      VDQN_LOAD_FRAGS(nxt_, tl_, cl_ & 1, nxt_) /* unconditional: the step behind the last one re-reads buffers that still exist */ \
      VDQN_MFMA_ALL(cur_)                                                                                                \
      VDQN_INTERLEAVE(8 + 2 * NF)                                                                                        \
    } else if constexpr ((FCNT_) == 1) { /* partial tile of a balanced walk, a wave with at most 16 rows of the range: one activation fragment, 8 MFMAs */ \
      VDQN_LOAD_FRAGS_N(nxt_, tl_, cl_ & 1, nxt_, 1)                                                                     \
      VDQN_MFMA_N(cur_, 1)                                                                                               \
      VDQN_INTERLEAVE(8)                                                                                                 \
    } /* FCNT_ 0: a wave without rows of the range only takes part in the staging and the barriers */                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }


  if constexpr (SK) {
    // ---- split-K remainder (round 5) ----
    // A launch of T tiles on `resident` workgroup slots costs ceil(T / resident) tile times although its last round only fills
    // T mod resident of the slots (3.06 rounds: 4 tile times).  The launcher therefore runs the whole rounds as one launch of the
    // kernel above and the r remaining tiles here: per XCD their r_x * cpk channel chunks (a chunk = nine K-steps) are dealt in
    // equal contiguous runs to the XCD's workgroups, so a workgroup computes one or two ITEMS = (tile, chunks [c0, c1)).  An item
    // that is a whole tile ends in the ordinary epilogue.  Otherwise the workgroup stores its f32 accumulators as part
    // (w - w_first) of the tile in the scratch slab and bumps the tile's arrival counter; the LAST workgroup to arrive adds the
    // parts IN PART ORDER (its own from the slab too: the sum does not depend on who arrives last) and runs the epilogue.
    // Nobody waits for anybody, so it does not matter which workgroups are resident at the same time.  All parts of a tile are
    // written and read inside one XCD; the counter operation is an agent-scope release / acquire all the same.
    const uint32_t xb = gridDim.x >> 3, w = blockIdx.x >> 3;  // (the grid is a multiple of 8)
    const uint32_t ucpk = (uint32_t)cpk;
    const uint32_t U = x_count * ucpk;
    const uint32_t u_beg = (uint32_t)(((unsigned long long)w * U) / xb), u_end = (uint32_t)(((unsigned long long)(w + 1) * U) / xb);
    // workgroup that holds chunk u of the XCD's run: the largest w' with floor(w' U / xb) <= u
    auto wg_of = [&](uint32_t u) { return (uint32_t)((((unsigned long long)(u + 1) * xb + U - 1) / U) - 1); };
    // (the arrival flag lives in weight buffer 0, free behind an item's K loop: a static __shared__ word on top of the 80 KB of
    // dynamic LDS would cost the second workgroup of the CU)
    volatile uint32_t* const sk_flag = reinterpret_cast<volatile uint32_t*>(smem);
    uint32_t u = u_beg;
    while (u < u_end) {
      const uint32_t j = u / ucpk;  // tile (index inside the XCD's remainder range)
      const int c0 = (int)(u - j * ucpk);
      const uint32_t t_end = (j + 1) * ucpk;
      const int c1 = (int)((u_end < t_end ? u_end : t_end) - j * ucpk);
      u = j * ucpk + (uint32_t)c1;
      const uint32_t tile = (uint32_t)bal_rows + x_first + j;
      tile_n = (int)(tile % (uint32_t)p.tiles_n);
      tile_m = (int)(tile / (uint32_t)p.tiles_n);
      n0 = tile_n * BN;
      m0 = tile_m * BM;
      q0 = m0 - W - 1 + lrow;
      b_off0 = (uint32_t)(n0 + lrow) * (uint32_t)(p.ktot * 2) + (uint32_t)(lchunk_b * 16);
      edge16 = edge_bits(m0);
      __syncthreads();  // the previous item's LDS reads (its epilogue's scratch, the flag) are done
      // the item's K-steps 0 and 1: window of chunk c0, weight tiles of taps 0 and 1
      VDQN_ISSUE_B(0, c0 * 128, b_off0, rs_b)
      VDQN_ISSUE_AW(0, c0 * 128, q0)
      VDQN_ISSUE_B(1, tap_k + c0 * 128, b_off0, rs_b)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf) acc[f][jf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      VDQN_LOAD_FRAGS(0, 0, 0, 0)
      const int n_ch = c1 - c0;
      {
        const uint32_t b_nx = b_off0;  // (behind a chunk pair: the item's next chunk — or chunks nobody reads)
        const i32x4 rs_bn = rs_b;
        const int q_nx = q0;
        // the 18-step body over chunk pairs; a run with an odd chunk count leaves it after its last chunk's nine steps (what steps
        // 7 and 8 staged ahead is read by nobody)
        _Pragma("clang loop unroll(disable)") for (int it = 0; 2 * it < n_ch; ++it) {
          const int cc2 = c0 + 2 * it;
          const int so_nx = (cc2 + 2) * 128;
          VDQN_USTEP(0, 4) VDQN_USTEP(1, 4) VDQN_USTEP(2, 4) VDQN_USTEP(3, 4) VDQN_USTEP(4, 4) VDQN_USTEP(5, 4)
          VDQN_USTEP(6, 4) VDQN_USTEP(7, 4) VDQN_USTEP(8, 4)
          if (2 * it + 1 >= n_ch) break;
          VDQN_USTEP(9, 4) VDQN_USTEP(10, 4) VDQN_USTEP(11, 4)
          VDQN_USTEP(12, 4) VDQN_USTEP(13, 4) VDQN_USTEP(14, 4) VDQN_USTEP(15, 4) VDQN_USTEP(16, 4) VDQN_USTEP(17, 4)
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      bool finish = true;
      if (c0 != 0 || c1 != cpk) {
        const uint32_t w_first = wg_of(j * ucpk), w_last = wg_of(t_end - 1);
        const uint32_t n_parts = w_last - w_first + 1, part = w - w_first;
        // part q of tile j sits in slot w_first + j + q of the XCD's 2 xb slots ((workgroup, tile) pairs are strictly ordered)
        float4* slab = reinterpret_cast<float4*>(p.sk_slab) + ((size_t)xcd * 2 * xb + w_first + j) * (size_t)(BM * BN / 4);
        // The hand-off follows MI355X_MICROARCH.md's fence-free form: EVERY store of a part is an sc1 store (written through to the
        // memory side, the line dropped from this XCD's L2), drained (vmcnt(0) per wave, then the workgroup barrier) before the
        // counter; EVERY load of a part is an sc1 load (never served by an L1 / a foreign L2).  An agent-scope release instead
        // writes back ALL dirty lines of the XCD's L2 — with 512 workgroups doing it at the end of a launch that cost more than
        // the split saved (profiles/r05v_bench_conv_splitk_fences.txt).
        // (scalar base + one 32-bit per-lane offset: sixteen 64-bit per-lane pointers would not fit beside the accumulators)
        const float4* const mine_u = slab + (size_t)part * (BM * BN / 4);
        const uint32_t lane_off = (uint32_t)tid * 16u;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int jf = 0; jf < NF; ++jf) {
            const uint32_t vo = lane_off + (uint32_t)((f * NF + jf) * G::NT * 16);
            asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(vo), "v"(acc[f][jf]), "s"(mine_u) : "memory");
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          unsigned* cnt = p.sk_cnt + xcd * 128u + j;
          const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == n_parts - 1) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (clean for the next launch)
          *sk_flag = old;
        }
        __syncthreads();
        finish = *sk_flag == n_parts - 1;
        if (finish) {
          const float4* src = slab;  // (uniform)
          for (uint32_t q = 0; q < n_parts; ++q) {
            // all sixteen loads of a part in flight before the one wait (the fragment registers are dead here): with four per wait
            // the eight round trips to the memory side cost a tile's last arriver ~12 us
            f32x4 v[4][NF];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
              for (int jf = 0; jf < NF; ++jf) {
                const uint32_t vo = lane_off + (uint32_t)((f * NF + jf) * G::NT * 16);
                asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v[f][jf]) : "v"(vo), "s"(src) : "memory");
              }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
              for (int jf = 0; jf < NF; ++jf) {
                asm volatile("" : "+v"(v[f][jf]));  // (the values arrive behind the wait above)
                acc[f][jf] = q == 0 ? v[f][jf] : acc[f][jf] + v[f][jf];
              }
            src += BM * BN / 4;
          }
        }
      }
      if (finish) {
        if (MODE == 0 && lean) lean_epilogue_128<WN>(le, acc, m0, n0, m_end, tid);
        else if (MODE == 1 && lean_d)
          lean_epilogue_dgrad_128<WN>(led, acc, reinterpret_cast<float*>(smem + kU_WinBase + kU_WinStride), m0, n0, tile_m, m_end, tid);
        else igemm_epilogue<T, BM, BN, MODE, WN>(p, acc, smem + kU_WinBase + kU_WinStride, m0, n0, tile_m, m_end, p.howo, W, 0, 0, p.bias);
      }
    }
    return;
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
    if constexpr (WALK == 2) break;  // (the partial tile: behind the loop)
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    VDQN_LOAD_FRAGS(0, 0, 0, 0)  // the fragments of step 0
    // the next tile of this workgroup (if any): where its first window and weight tiles come from
    // (balanced walk: the next 128 rows of the workgroup's range, same column tile, same weights)
    const uint32_t lt_nx = lt + x_blocks;
    const bool has_nx = BAL ? m0 + BM < m_end : lt_nx < x_count;
    const int tn_nx = (BAL || !has_nx) ? tile_n : (int)(((uint32_t)bal_rows + x_first + lt_nx) % (uint32_t)p.tiles_n);
    const int tm_nx = (BAL || !has_nx) ? tile_m : (int)(((uint32_t)bal_rows + x_first + lt_nx) / (uint32_t)p.tiles_n);
    const int m0_nx = BAL ? (has_nx ? m0 + BM : m0) : tm_nx * BM;
    const int q0_t = BAL ? 0 : m0_nx - W - 1 + lrow;
    const uint32_t b_t = BAL ? b_off0 : (uint32_t)(tn_nx * BN + lrow) * (uint32_t)(p.ktot * 2) + (uint32_t)(lchunk_b * 16);
    const i32x4 rs_t = m0_nx >= m_split ? rs_b1 : rs_b0;  // the next tile's weight set
#define VDQN_ULOOP(FCNT_, NEXT_)                                                                                          \
  _Pragma("clang loop unroll(disable)") for (int it = 0; it < n_it; ++it) {                                              \
    const int cc2 = 2 * it; /* first chunk of this iteration */                                                          \
    /* (without a next tile the steps behind the end stage this tile's chunk 2 it + 2: nobody reads it) */               \
    const bool last_it = (NEXT_) && has_nx && it == n_it - 1;                                                            \
    const int so_nx = last_it ? 0 : (cc2 + 2) * 128;                                                                     \
    const int q_nx = BAL ? q0 + (last_it ? BM : 0) : (last_it ? q0_t : q0); /* (balanced walk: the next 128 rows) */        \
    const uint32_t b_nx = last_it ? b_t : b_off0;                                                                        \
    const i32x4 rs_bn = last_it ? rs_t : rs_b;                                                                           \
    VDQN_USTEP(0, FCNT_) VDQN_USTEP(1, FCNT_) VDQN_USTEP(2, FCNT_) VDQN_USTEP(3, FCNT_) VDQN_USTEP(4, FCNT_) VDQN_USTEP(5, FCNT_)       \
    VDQN_USTEP(6, FCNT_) VDQN_USTEP(7, FCNT_) VDQN_USTEP(8, FCNT_) VDQN_USTEP(9, FCNT_) VDQN_USTEP(10, FCNT_) VDQN_USTEP(11, FCNT_)     \
    VDQN_USTEP(12, FCNT_) VDQN_USTEP(13, FCNT_) VDQN_USTEP(14, FCNT_) VDQN_USTEP(15, FCNT_) VDQN_USTEP(16, FCNT_) VDQN_USTEP(17, FCNT_) \
  }
    VDQN_ULOOP(4, true)
    VDQN_ST(st_comp)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two tiles staged behind the last step have landed (the next tile's
    __builtin_amdgcn_s_barrier();                     // K-steps 0 and 1: weight buffers 0, 1 and window buffer 0 stay untouched)
#ifdef VDQN_STAMP
    st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    // the epilogue's scratch (column sums) is window buffer 1: its last reader was step 16.  (m_end: rows behind the workgroup's
    // range — or behind M — are not stored)
    if constexpr (BAL) {  // (the launcher takes the balanced walk only for calls the lean epilogue serves)
      lean_epilogue_128<WN>(le, acc, m0, n0, m_end, tid);
    } else {
      if (MODE == 0 && lean) lean_epilogue_128<WN>(le, acc, m0, n0, m_end, tid);
      else if (MODE == 1 && lean_d)
        lean_epilogue_dgrad_128<WN, BM / 64>(led, acc, reinterpret_cast<float*>(smem + kU_WinBase + kU_WinStride), m0, n0, tile_m, m_end, tid);
      else igemm_epilogue<T, BM, BN, MODE, WN>(p, acc, smem + kU_WinBase + kU_WinStride, m0, n0, tile_m, m_end, p.howo, W, 0, 0,
                                               m0 >= m_split ? p.bias_b : p.bias);
    }
    if (!has_nx) {
      m0 = m_end;  // (balanced walk: nothing is left for the section behind the loop)
      break;
    }
    lt = lt_nx;
    tile_n = tn_nx; tile_m = BAL ? m0_nx / BM : tm_nx;
    n0 = tile_n * BN; m0 = m0_nx;
    q0 = BAL ? q0 + BM : q0_t;
    b_off0 = b_t;
    rs_b = rs_t;
    edge16 = edge_bits(m0);
  }
  if constexpr (WALK == 2) {
    // The partial last tile of a row range (m_end - m0 < 128 rows; its first two K-steps were staged by the prologue).  A wave
    // computes all four of its 16-row fragments, ONE, or none — whichever hold rows of the range — in its own copy of the K loop;
    // every copy runs the same barriers and DMA pieces.  Nothing follows, so nothing is staged ahead.
    const int rows_wave = m_end - m0 - (wave_u / WN) * 64;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool has_nx = false;
    const int q0_t = q0;
    const uint32_t b_t = b_off0;
    const i32x4 rs_t = rs_b;
    if (rows_wave > 16) {
      VDQN_LOAD_FRAGS(0, 0, 0, 0)
      VDQN_ULOOP(4, false)
    } else if (rows_wave > 0) {
      VDQN_LOAD_FRAGS_N(0, 0, 0, 0, 1)
      VDQN_ULOOP(1, false)
    } else {
      VDQN_ULOOP(0, false)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    lean_epilogue_128<WN>(le, acc, m0, n0, m_end, tid);
  }
#undef VDQN_ULOOP
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
static int g_win9_splitk_override = -1;
extern "C" void vdqn_debug_set_win9_splitk(int v) { g_win9_splitk_override = v; }  // test hook (not part of include/vdqn.h): -1 = VDQN_WIN9_SPLITK
static int splitk_env() {
  static const int v = [] { const char* e = getenv("VDQN_WIN9_SPLITK"); return e ? atoi(e) : 0; }();
  return v;
}
// bytes of vdqn_conv_args.splitk_ws that serve any call: 4 KiB of arrival counters + two 128 x 128 f32 parts per resident workgroup
extern "C" int64_t vdqn_conv2d_splitk_workspace_bytes(void) { return (int64_t)4096 + (int64_t)2 * 2 * vdqn_num_cus() * (128 * 128 * 4); }
static int g_win9_balanced_override = -1;
extern "C" void vdqn_debug_set_win9_balanced(int v) { g_win9_balanced_override = v; }  // test hook (not part of include/vdqn.h)

// entry used by vdqn_conv2d (igemm.hip): returns VDQN_OK or an error code
// tile_first / tile_count: the part of the launch's tile sequence (row block major, column tile fastest, in tiles of BM rows) this call
// covers; tile_count < 0 = all of it
template <int MODE, int BM>
static void launch_win9u(const IgemmParams& p, hipStream_t stream, void* stamps, int tile_first = 0, int tile_count = -1) {
  using G = Win9Geom<BM>;
  const int wrows = (BM + 2 * p.wo + 2 + 1 + 7) & ~7;  // <= G::WinRows for W <= 28
  const bool part = tile_count >= 0;
  const unsigned tiles = part ? (unsigned)tile_count : (unsigned)(((p.M + BM - 1) / BM) * p.tiles_n);
  // VDQN_WIN9_PERSIST=1: at most as many workgroups as the chip holds at once, each walking its tiles with the next tile's first
  // two K-steps staged under the current tile's last steps and epilogue; 0: one workgroup per tile
  // (default: for launches of more than two rounds of resident workgroups — layer2 +5-7 %, layer3 at 512 frames +4 %; a launch of
  // 1.5 rounds loses 2-3 % to the static tile assignment; 2: always; profiles/r02o_win9u_persistent.txt)
  static const int persist = [] { const char* e = getenv("VDQN_WIN9_PERSIST"); return e ? atoi(e) : 1; }();
  // VDQN_WIN9_BALANCED (round 5; default 0 = off): forward launches of more than one round of resident workgroups, without column
  // sums and not grouped, split their rows EVENLY over the resident workgroups of each column tile (the kernel's balanced walk:
  // full tiles in one launch, every range's partial last tile in a second) instead of dealing whole tiles.  Built against the
  // partial rounds of the static walk (3.06 rounds cost 4 tile times) and MEASURED SLOWER: a partial tile with 1/16 of the MFMA work
  // still takes 1.3-1.45 full tile times — the K-step is bound by its staging (18.7 KB of LDS-DMA per workgroup and step, one step of
  // lead), not by the matrix work: layer3 at 512 frames 120.0 vs 107.6 us, layer2 160.6 vs 148.5 (profiles/r05d_bench_win9_*.txt,
  // DESIGN.md section 6d).  1: on for launches of more than one round; 2: also smaller launches.  vdqn_debug_set_win9_balanced
  // overrides the environment (tests).
  static const int balanced_env = [] { const char* e = getenv("VDQN_WIN9_BALANCED"); return e ? atoi(e) : 0; }();
  const int balanced = g_win9_balanced_override >= 0 ? g_win9_balanced_override : balanced_env;
  const unsigned resident = (unsigned)((BM == 128 ? 2 : 1) * vdqn_num_cus());
  int bal_rows = 0;
  // (256-row tiles, one workgroup per CU: persistent above one round — VDQN_WIN9_PERSIST256=0: one workgroup per tile, as before round 5)
  static const int persist256 = [] { const char* e = getenv("VDQN_WIN9_PERSIST256"); return e ? atoi(e) : 1; }();
  unsigned grid = ((BM == 128 || persist256) && ((persist == 1 && tiles > 2 * resident) || (persist >= 2 && tiles > resident) || (BM == 256 && persist256 && tiles > resident))) ? resident : tiles;
  const bool lean_ok = !p.no_lean && p.co % 128 == 0 && p.vec_ok && p.out && !p.out_f32 && !p.mask && p.bias && (((uintptr_t)p.bias) & 15) == 0 && (long long)p.M * p.ldo * 2 < 0x7fffffffLL;
  if (BM == 128 && MODE == 0 && balanced && !part && lean_ok && !p.colsum_part && !p.wt_b && (tiles > resident || balanced >= 2) && p.tiles_n > 0 &&
      resident % (8u * (unsigned)p.tiles_n) == 0 && resident / (unsigned)p.tiles_n >= 8u) {
    const unsigned per_col = resident / (unsigned)p.tiles_n;       // workgroups (= row ranges) per column tile
    bal_rows = (int)((((long long)p.M + per_col - 1) / per_col + 15) / 16 * 16);
    grid = resident;
  }
  if constexpr (BM == 128 && MODE == 0) {
    if (bal_rows > 0) {
      vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9u_kernel<MODE, BM, 1>), (size_t)G::Smem);
      vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9u_kernel<MODE, BM, 2>), (size_t)G::Smem);
      if (bal_rows >= BM)
        hipLaunchKernelGGL((win9u_kernel<MODE, BM, 1>), dim3(grid), dim3(G::NT), G::Smem, stream, p, wrows, make_fastdiv((uint32_t)p.wo),
                           make_fastdiv((uint32_t)p.howo), tiles, stamps, bal_rows);
      if (bal_rows % BM != 0 || p.M % bal_rows != 0)  // some range ends with a partial tile
        hipLaunchKernelGGL((win9u_kernel<MODE, BM, 2>), dim3(grid), dim3(G::NT), G::Smem, stream, p, wrows, make_fastdiv((uint32_t)p.wo),
                           make_fastdiv((uint32_t)p.howo), tiles, stamps, bal_rows);
      return;
    }
  }
  // Split-K remainder (VDQN_WIN9_SPLITK=1; default 0 = off; needs vdqn_conv_args.splitk_ws): the launch's whole rounds of `resident`
  // tiles run as before, the r tiles behind them as a second launch whose workgroups each take an equal run of the remainder's
  // channel chunks (win9u_kernel<.., 3>).  Taken when the longest run (+ ~6 K-steps for the second launch, the items' own
  // prologues and the reduction of the parts) is shorter than the tile it replaces; 2: whenever there is a whole round and a
  // remainder.  Built against the partial last rounds of the static walk and MEASURED SLOWER (rocprofv3 kernel trace,
  // profiles/r05w_splitk_kernel_trace.txt): the last round of the unsplit launch is cheap already — its few workgroups run alone on
  // their CUs at about twice the K-step rate of a full chip (layer3 at 512 frames: 4 rounds in 110 us where 3 whole rounds take
  // 96.5) — while the remainder launch pays ~7 us of start-up, hand-off (64 KB of sc1 stores per part, a counter, the last
  // arriver's reads) and epilogue on top of its K-steps: 15.6 us against ~13.5 for layer3 at 512 frames, 62.5 against ~53 for
  // layer4 at 512 (272 remainder tiles: every workgroup holds two partial items), 49.6 against ~30 for layer3's data gradient.
  if constexpr (BM == 128) {
    const int splitk = g_win9_splitk_override >= 0 ? g_win9_splitk_override : splitk_env();
    const unsigned whole = tiles / resident * resident, rem = tiles - whole;
    const int cpk = p.ci / 64;
    if (splitk && !part && p.sk_cnt && p.sk_slab && !p.wt_b && whole > 0 && rem > 0 && resident % 8u == 0 && resident / 8u <= 64u) {
      const unsigned run = (rem * (unsigned)cpk + resident - 1) / resident;  // chunks of the longest run
      if (splitk >= 2 || 9 * run + 6 < 9u * (unsigned)cpk + 3) {
        const unsigned grid1 = ((persist == 1 && whole > 2 * resident) || (persist >= 2 && whole > resident)) ? resident : whole;
        vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9u_kernel<MODE, BM>), (size_t)G::Smem);
        vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9u_kernel<MODE, BM, 3>), (size_t)G::Smem);
        hipLaunchKernelGGL((win9u_kernel<MODE, BM>), dim3(grid1), dim3(G::NT), G::Smem, stream, p, wrows, make_fastdiv((uint32_t)p.wo),
                           make_fastdiv((uint32_t)p.howo), whole, stamps, 0);
        hipLaunchKernelGGL((win9u_kernel<MODE, BM, 3>), dim3(resident), dim3(G::NT), G::Smem, stream, p, wrows, make_fastdiv((uint32_t)p.wo),
                           make_fastdiv((uint32_t)p.howo), rem, stamps, (int)whole);
        return;
      }
    }
  }
  IgemmParams q = p;
  q.sk_cnt = nullptr;  // (no remainder launch behind this one: nothing to clear)
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9u_kernel<MODE, BM>), (size_t)G::Smem);
  hipLaunchKernelGGL((win9u_kernel<MODE, BM>), dim3(grid), dim3(G::NT), G::Smem, stream, q, wrows, make_fastdiv((uint32_t)p.wo),
                     make_fastdiv((uint32_t)p.howo), tiles, stamps, tile_first);
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
  // MIXED tiling (VDQN_WIN9_MIXED=1; default 0): where the fractional part is small, the launch's WHOLE rounds on 256-row tiles (the
  // faster K-step) and only the rows behind them on 128-row tiles, whose partial round is the cheap one — two launches over disjoint
  // rows, every output element computed by the same K order as before (bit-identical).  MEASURED SLOWER: layer3 at 512 frames 116.1
  // vs 105.8 us, at 384 frames 89.7 vs 79.2, layer4 at 384 frames 99.7 vs 92.7; in the update the forward launches 1.637 vs 1.606 ms
  // (profiles/r06i_*): the second launch starts only when the first has drained and pays a launch gap of its own, which costs more
  // than the 6 % the 256-row rounds save.
  static const int mixed_env = [] { const char* e = getenv("VDQN_WIN9_MIXED"); return e ? atoi(e) : 0; }();
  int mixed_tiles128 = 0;  // > 0: 128-row tiles [0, mixed_tiles128) run as 256-row tiles, the rest as they are
  if (bm256 == 3 && p.ci >= 256 && !p.wt_b) {
    const long long tiles128 = (long long)p.tiles_m * p.tiles_n, res128 = 2ll * vdqn_num_cus();
    const double r = (double)tiles128 / (double)res128;  // rounds of the 128-row tiles (p.tiles_m counts those)
    const double frac = r - (double)(long long)r;
    big = r >= 0.7 && frac >= 0.45;
    const long long whole = tiles128 / res128 * res128;
    if (!big && mixed_env && whole > 0 && whole < tiles128 && whole % (2ll * p.tiles_n) == 0) mixed_tiles128 = (int)whole;
  }
  void* stamps = nullptr;
#ifdef VDQN_STAMP
  stamps = g_stamp_buffer;
#endif
  vdqn_prof_begin(mode == 0 ? "igemm_win<bf16,128,fwd>" : "igemm_win<bf16,128,dgrad>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))), stream);
  if (mixed_tiles128 > 0) {
    const int n128 = p.tiles_m * p.tiles_n - mixed_tiles128;
    if (mode == 0) {
      launch_win9u<0, 256>(p, stream, stamps, 0, mixed_tiles128 / 2);
      launch_win9u<0, 128>(p, stream, stamps, mixed_tiles128, n128);
    } else {
      launch_win9u<1, 256>(p, stream, stamps, 0, mixed_tiles128 / 2);
      launch_win9u<1, 128>(p, stream, stamps, mixed_tiles128, n128);
    }
  } else if (mode == 0) {
    if (big) launch_win9u<0, 256>(p, stream, stamps); else launch_win9u<0, 128>(p, stream, stamps);
  } else {
    if (big) launch_win9u<1, 256>(p, stream, stamps); else launch_win9u<1, 128>(p, stream, stamps);
  }
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
