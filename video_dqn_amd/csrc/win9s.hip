// Plane-window kernel for the 3x3 / stride-2 / pad-1 convolutions (forward, bf16): conv1 of the first BasicBlock of ResNet layer2,
// layer3 and layer4 (torchvision resnet.py, reached from archs/HabitatDQNMultiAction.py:30,49-51).  On the generic implicit GEMM these
// three layers run at ~570 TFLOP/s against ~1000 for the stride-1 window kernels: every one of the nine taps re-stages its own
// 128-row activation tile by a per-row gather (3.4 vector + 2 scalar instructions per MFMA, SQ_INSTS_VALU / SQ_INSTS_MFMA in
// profiles/r03d_pmc_mfma.json).
//
// Decomposition.  With an even H x W input, output pixel (oh, ow) reads input rows 2 oh + kr - 1 and columns 2 ow + ks - 1.  Split
// the input into its four parity planes P_ab[y][x] = in[2 y + a][2 x + b] — each as large as the OUTPUT image — and the convolution is
// four stride-ONE convolutions with taps that only reach up and left:
//     kr = 0 -> plane row parity a = 1, dy = -1;   kr = 1 -> a = 0, dy = 0;   kr = 2 -> a = 1, dy = 0     (columns alike with ks, b, dx)
//     P11: taps (0,0) (0,2) (2,0) (2,2) at (dy, dx) = (-1,-1) (-1,0) (0,-1) (0,0);  P10: (0,1) (2,1) at (-1,0) (0,0);
//     P01: (1,0) (1,2) at (0,-1) (0,0);  P00: (1,1) at (0,0).
// So, as in win9.hip, ONE staged window per (plane, 64-channel chunk) — 128 + Wo + 1 consecutive plane pixels — serves all taps of
// that plane (tile row r reads window row r + Wo + 1 + dy Wo + dx; a lane whose tap leaves the image — top row with dy = -1, left
// column with dx = -1 — reads the zero pair at its own bank position): four windows per nine K-steps instead of nine tiles.  The
// planes are never materialised: the LDS-DMA's source address is per lane, window row j of plane (a, b) is fetched from input pixel
// (img, 2 y + a, 2 x + b); the per-row part of that address is computed once per tile (two divisions per staged row), the plane and
// the channel chunk are a scalar offset.
//
// Everything else is win9u_kernel's: 128 x 128 tiles, 4 waves of 64 x 64, weight tiles streamed per K-step (same packed operand
// [co][kr][ks][ci]), two LDS buffers for weights and for windows, two register sets of fragments, one barrier per K-step, the K loop
// unrolled (18 steps = two channel chunks, plus a 9-step block for an odd chunk count: layer2.0 has ONE chunk), two workgroups per
// CU, the shared epilogue.  One workgroup per tile.
#include <stdlib.h>

#include "igemm_common.h"

namespace {

constexpr int kS_WtTile = 128 * 128;        // one staged weight tile
constexpr int kS_WinRows = 160;             // >= 128 + 28 + 1 (+ the zero pair), a multiple of the 32-row staging pass
constexpr int kS_WinStride = kS_WinRows * 128;
constexpr int kS_WinBase = 2 * kS_WtTile;   // LDS: [2 weight tiles][2 windows]
constexpr int kS_Smem = kS_WinBase + 2 * kS_WinStride;
constexpr int kS_WPass = kS_WinRows / 32;   // 5 staging passes per window
constexpr int kSP_BiasCols = 512;            // win9sp_kernel: the layer's biases (3x3 and sibling) live in LDS behind the windows
constexpr int kSP_Smem = kS_Smem + 2 * kSP_BiasCols * 4;

// step s (0..8) of a chunk: tap index kr * 3 + ks, plane (a, b), shift class (dy, dx), window buffer
struct S2Step { int tap, a, b, dy, dx, wbuf; };
__host__ __device__ constexpr S2Step s2_step(int s) {
  constexpr S2Step tab[9] = {{0, 1, 1, -1, -1, 0}, {2, 1, 1, -1, 0, 0}, {6, 1, 1, 0, -1, 0}, {8, 1, 1, 0, 0, 0},  // P11 -> window buffer 0
                             {1, 1, 0, -1, 0, 1},  {7, 1, 0, 0, 0, 1},                                            // P10 -> 1
                             {3, 0, 1, 0, -1, 0},  {5, 0, 1, 0, 0, 0},                                            // P01 -> 0
                             {4, 0, 0, 0, 0, 1}};                                                                 // P00 -> 1
  return tab[s];
}
// the window a step needs is staged two steps earlier, when its buffer's last reader (the fragment reads of the step before that)
// is done: step s issues the window whose first user is step s + 2 — steps 2, 4, 6 issue P10, P01, P00 of the same chunk, step 7
// issues P11 of the NEXT chunk
__host__ __device__ constexpr int s2_window_issued_at(int s) { return s == 2 ? 4 : s == 4 ? 6 : s == 6 ? 8 : s == 7 ? 9 : -1; }

__global__ __launch_bounds__(256, 2) void win9s_kernel(const IgemmParams p, const FastDiv d_wo, const FastDiv d_howo) {
  using T = bf16raw;
  constexpr int BM = 128, BN = 128, WN = 2, NF = 4, CPL = 16;
  constexpr int PSTR = 32 * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lb % (uint32_t)p.tiles_n), tile_m = (int)(lb / (uint32_t)p.tiles_n);
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int Wo = p.wo, rows_total = p.M;
  const int lrow = tid >> 3;
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int lchunk_b = (tid & 7) ^ ((((lrow / CPL) & 1) << 2) | (lrow & 3));

  const unsigned long long a_ptr = (unsigned long long)p.in, b_ptr = (unsigned long long)p.wt;
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};

  // ---- window rows staged by this thread: j = lrow + 32 i <-> plane pixel q = m0 - Wo - 1 + j = (img, y, x).  Its input pixel for
  // plane (a, b) is (img, 2 y + a, 2 x + b): the (0, 0) pixel's byte offset here, once per tile; rows past 128 + Wo + 1 and pixels
  // outside the tensor get an out-of-range offset (zero fill; the last two rows are the zero pair) ----
  const int pixB = p.pix_stride * 2;
  const int need = BM + Wo + 1;
  uint32_t a_row[kS_WPass];
#pragma unroll
  for (int i = 0; i < kS_WPass; ++i) {
    const int j = lrow + 32 * i;
    const int q = m0 - Wo - 1 + j;
    const bool ok = j < need && (unsigned)q < (unsigned)rows_total;
    const uint32_t qq = ok ? (uint32_t)q : 0u;
    const uint32_t img = fastdiv(qq, d_howo), rem = qq - img * d_howo.div;
    const uint32_t y = fastdiv(rem, d_wo), x = rem - y * d_wo.div;
    a_row[i] = ok ? ((img * (uint32_t)p.hi + 2u * y) * (uint32_t)p.wi + 2u * x) * (uint32_t)pixB + (uint32_t)(lchunk_a * 16) : kOob;
  }
  const uint32_t b_off0 = (uint32_t)(n0 + lrow) * (uint32_t)(p.ktot * 2) + (uint32_t)(lchunk_b * 16);
  const int b_row32 = 32 * p.ktot * 2;

  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * (8 * 128);

  // window of plane (A_, B_), channel chunk CC -> window buffer WBUF: five pieces (scalar offset = plane pixel + chunk)
#define VDQN_S_ISSUE_AW(WBUF, A_, B_, CC)                                                                          \
  {                                                                                                                \
    const uint32_t la_ = lds_wave + (uint32_t)(kS_WinBase + (WBUF)*kS_WinStride);                                  \
    const int so_ = ((A_)*p.wi + (B_)) * pixB + (CC)*128;                                                          \
    asm volatile(                                                                                                  \
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %6, %7 offen lds\n\t"                             \
        "s_add_u32 m0, %5, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %6, %7 offen lds\n\t"                         \
        "s_add_u32 m0, %5, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %7 offen lds\n\t"                         \
        "s_add_u32 m0, %5, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %7 offen lds\n\t"                        \
        "s_add_u32 m0, %5, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %7 offen lds"                             \
        ::"v"(a_row[0]), "v"(a_row[1]), "v"(a_row[2]), "v"(a_row[3]), "v"(a_row[4]), "s"(la_), "s"(rs_a), "s"(so_),  \
          "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR), "n"(4 * PSTR)                                                   \
        : "memory", "scc");                                                                                        \
  }
  // weight tile of tap TAP_, chunk CC -> weight buffer BUF: four pieces (rows lrow + 32 i of the column tile)
#define VDQN_S_ISSUE_B(BUF, TAP_, CC)                                                                              \
  {                                                                                                                \
    const uint32_t lb_ = lds_wave + (uint32_t)((BUF)*kS_WtTile);                                                   \
    const int so0_ = (TAP_)*tap_k + (CC)*128, so1_ = so0_ + b_row32, so2_ = so1_ + b_row32, so3_ = so2_ + b_row32; \
    asm volatile(                                                                                                  \
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\t"                             \
        "s_add_u32 m0, %1, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %4 offen lds\n\t"                         \
        "s_add_u32 m0, %1, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %5 offen lds\n\t"                         \
        "s_add_u32 m0, %1, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %6 offen lds"                              \
        ::"v"(b_off0), "s"(lb_), "s"(rs_b), "s"(so0_), "s"(so1_), "s"(so2_), "s"(so3_), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
        : "memory", "scc");                                                                                        \
  }

  f32x4 acc[4][NF];
  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  // edge bits of this lane's four pixels, 2 bits per fragment f: 1 top row (dy = -1 leaves the image), 2 left column (dx = -1)
  uint32_t edge8 = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const uint32_t m = (uint32_t)(m0 + wr * 64 + f * 16 + i16);
    const uint32_t rem = m - fastdiv(m, d_howo) * d_howo.div;
    const uint32_t oh = fastdiv(rem, d_wo), ow = rem - oh * d_wo.div;
    edge8 |= ((oh == 0 ? 1u : 0u) | (ow == 0 ? 2u : 0u)) << (2 * f);
  }
  // per-lane LDS offsets of the four shift classes (inside a window buffer): tile row wr*64 + i16 reads window row r + Wo + 1 + dy Wo + dx
  uint32_t ab[4];  // [2 * (dy + 1) + (dx + 1)]
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int joff = Wo + 1 - ((c & 2) ? 0 : Wo) - ((c & 1) ? 0 : 1);
    const int row = wr * 64 + i16 + joff;
    ab[c] = (uint32_t)(row * 128 + ((g ^ ((i16 + joff) & 7)) << 4));
  }
  uint32_t zs[4];  // the zero pair (rows kS_WinRows - 2, - 1: staged as out-of-range rows), minus the f * 16 rows the read's immediate adds
#pragma unroll
  for (int f = 0; f < 4; ++f) zs[f] = (uint32_t)((kS_WinRows - 2) * 128 - f * 16 * 128);
  const uint32_t bb0 = (uint32_t)((wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128 + ((g ^ (i16 & 7)) << 4));
  const uint32_t bb1 = (uint32_t)((wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128 + (((g + 4) ^ (i16 & 7)) << 4));

  const int cpk = p.ci / 64;      // channel chunks
  const int tap_k = cpk * 128;    // byte distance between the weight K offsets of consecutive taps of one chunk

  u32x4 fa[2][2][4], fb[2][2][NF];  // [register set][K half][fragment]

  // fragments of step S_ (0..8 of a chunk; window buffer and shift class from the table), weight buffer BBUF_ -> register set SET
#define VDQN_S_LOAD_FRAGS(SET, S_, BBUF_)                                                                          \
  {                                                                                                                \
    constexpr S2Step st_ = s2_step(S_);                                                                            \
    constexpr uint32_t tb_ = (st_.dy < 0 ? 1u : 0u) | (st_.dx < 0 ? 2u : 0u);                                      \
    const unsigned char* wb_ = smem + kS_WinBase + st_.wbuf * kS_WinStride;                                        \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                             \
      uint32_t a0_ = ab[2 * (st_.dy + 1) + (st_.dx + 1)];                                                          \
      if constexpr (tb_ != 0u) {                                                                                   \
        const bool z_ = (edge8 & (tb_ << (2 * f_))) != 0u;                                                         \
        a0_ = z_ ? ((a0_ & 255u) | zs[f_]) : a0_;                                                                  \
      }                                                                                                            \
      const uint32_t a1_ = a0_ ^ 64u;                                                                              \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a0_);                                 \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a1_);                                 \
    }                                                                                                              \
    const unsigned char* bt_ = smem + (BBUF_)*kS_WtTile;                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                            \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + bb0);                                  \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + bb1);                                  \
    }                                                                                                              \
  }
#define VDQN_S_MFMA_ALL(SET)                                                                                       \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)               \
      _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                          \
    acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[SET][h_][j_]),             \
                                                          __builtin_bit_cast(bf16x8, fa[SET][h_][f_]), acc[f_][j_], 0, 0, 0); \
  }
  // K-step U of a block that starts at chunk C0_ (U = 0..17 over two chunks, or 0..8 over one): its fragments are in register set
  // U & 1; it issues the weight tile of step U + 2 into the weight buffer it has just released and, at the table's steps, the
  // window whose first user is two steps ahead; it reads the fragments of step U + 1 underneath its own MFMAs
#define VDQN_S_USTEP(U, C0_)                                                                                       \
  {                                                                                                                \
    constexpr int cur_ = (U)&1, nxt_ = cur_ ^ 1;                                                                   \
    constexpr int s2_ = ((U) + 2) % 9, c2_ = ((U) + 2) / 9; /* step and chunk (relative to C0_) staged now */       \
    constexpr int s1_ = ((U) + 1) % 9;                      /* ... whose fragments are read now */                 \
    constexpr int wi_ = s2_window_issued_at((U) % 9);      /* first user of the window issued in this step, or -1 */ \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
    asm volatile("" : "+v"(edge8));                                                                                \
    asm volatile("" : "+v"(fa[cur_][0][0]), "+v"(fa[cur_][0][1]), "+v"(fa[cur_][0][2]), "+v"(fa[cur_][0][3]),      \
                      "+v"(fa[cur_][1][0]), "+v"(fa[cur_][1][1]), "+v"(fa[cur_][1][2]), "+v"(fa[cur_][1][3]));     \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[cur_][0][j_]), "+v"(fb[cur_][1][j_])); \
    __builtin_amdgcn_s_barrier();                                                                                  \
    /* (behind the last chunk these stage tiles nobody reads: offsets past a row's channels fetch a neighbour's bytes, in range) */ \
    VDQN_S_ISSUE_B(cur_, s2_step(s2_).tap, (C0_) + c2_)                                                            \
    if constexpr (wi_ >= 0) VDQN_S_ISSUE_AW(s2_step(wi_ % 9).wbuf, s2_step(wi_ % 9).a, s2_step(wi_ % 9).b, (C0_) + ((U) / 9) + wi_ / 9) \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    VDQN_S_LOAD_FRAGS(nxt_, s1_, nxt_)                                                                             \
    VDQN_S_MFMA_ALL(cur_)                                                                                          \
    VDQN_INTERLEAVE(8 + 2 * NF)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }

  // prologue: K-steps 0 and 1 of chunk 0 (P11's window, weight tiles of taps 0 and 2)
  VDQN_S_ISSUE_B(0, s2_step(0).tap, 0)
  VDQN_S_ISSUE_AW(0, 1, 1, 0)
  VDQN_S_ISSUE_B(1, s2_step(1).tap, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  VDQN_S_LOAD_FRAGS(0, 0, 0)  // the fragments of step 0
  const int n_it = cpk >> 1;
#pragma clang loop unroll(disable)
  for (int it = 0; it < n_it; ++it) {
    const int c0 = 2 * it;
    VDQN_S_USTEP(0, c0) VDQN_S_USTEP(1, c0) VDQN_S_USTEP(2, c0) VDQN_S_USTEP(3, c0) VDQN_S_USTEP(4, c0) VDQN_S_USTEP(5, c0)
    VDQN_S_USTEP(6, c0) VDQN_S_USTEP(7, c0) VDQN_S_USTEP(8, c0) VDQN_S_USTEP(9, c0) VDQN_S_USTEP(10, c0) VDQN_S_USTEP(11, c0)
    VDQN_S_USTEP(12, c0) VDQN_S_USTEP(13, c0) VDQN_S_USTEP(14, c0) VDQN_S_USTEP(15, c0) VDQN_S_USTEP(16, c0) VDQN_S_USTEP(17, c0)
  }
  if (cpk & 1) {  // an odd chunk count (layer2.0: one chunk): the last chunk as a 9-step block (18 n_it steps lie behind: set 0 again)
    const int c0 = cpk - 1;
    VDQN_S_USTEP(0, c0) VDQN_S_USTEP(1, c0) VDQN_S_USTEP(2, c0) VDQN_S_USTEP(3, c0) VDQN_S_USTEP(4, c0) VDQN_S_USTEP(5, c0)
    VDQN_S_USTEP(6, c0) VDQN_S_USTEP(7, c0) VDQN_S_USTEP(8, c0)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the tiles staged behind the last step have landed; the reads too
  __builtin_amdgcn_s_barrier();
  igemm_epilogue<T, BM, BN, 0, WN>(p, acc, smem + kS_WinBase, m0, n0, tile_m, rows_total, p.howo, Wo, 0, 0, p.bias);
#undef VDQN_S_USTEP
#undef VDQN_S_LOAD_FRAGS
#undef VDQN_S_MFMA_ALL
#undef VDQN_S_ISSUE_AW
#undef VDQN_S_ISSUE_B
}


// ---------------------------------------------------------------------------------------------------------
// Round 5: the PERSISTENT form, with the block's 1x1 / stride-2 downsample computed in the same launch (SIB).
//
// A layer2.0 tile is 9 K-steps, a layer3.0 tile 18: with one workgroup per tile the prologue (three DMA groups, a full wait, a
// barrier) and the epilogue are a large share of a tile (PMC: 43 % of the wave-cycles parked at waitcnt / barrier,
// profiles/r04bf_pmc_mfma.json).  Here a launch has at most as many workgroups as the chip holds at once (two per CU); each walks
// its XCD's tiles (win9u_kernel's walk) and the last two K-steps of a tile stage the first two K-steps of what comes next, so the
// next prologue runs under this tile's last MFMAs and its epilogue.
//
// The downsample (torchvision BasicBlock.downsample = conv1x1(stride 2) + BatchNorm, reached from
// archs/HabitatDQNMultiAction.py:30) reads input pixel (2 oh, 2 ow) = plane P00 at shift (0, 0): exactly the activation
// fragments of the centre tap.  A second accumulator set does not fit beside two fragment sets (64 + 64 + 128 registers of 256),
// so the 1x1 runs as CPK extra K-steps BEHIND the 3x3's epilogue, on the same accumulators: [9 CPK steps][epilogue -> out]
// [CPK steps: P00 window of chunk d, weight tile d of wt2][epilogue -> out2], each boundary pipelined like a tile boundary.  Its K
// order is chunk 0, 1, .. as in its own launch on the generic kernel: out2 is bit-identical to that launch.
// Window buffers: the 3x3 steps keep the table's assignment; downsample step d uses buffer d & 1 (one chunk: the P00 window of
// step 8 is still in buffer 1).  Every window is staged two steps ahead of its first reader into the buffer whose last reader is at
// least two steps back — checked case by case for CPK = 1, 2, 4 in DESIGN.md section 3d.
// The step count per tile must be even (two weight buffers / two register sets alternate by step parity): 9 CPK + CPK always is,
// 9 CPK alone needs an even CPK; the one-chunk layer without a sibling keeps win9s_kernel.
// ---------------------------------------------------------------------------------------------------------
template <int CPK, bool SIB>
__global__ __launch_bounds__(256, 2) void win9sp_kernel(const IgemmParams p, const FastDiv d_wo, const FastDiv d_howo, const uint32_t total_tiles, const int tiles_n,
                                                       void* stamps) {
  static_assert(CPK == 1 || CPK == 2 || CPK == 4, "channel chunks of layer2.0 / layer3.0 / layer4.0");
  static_assert(SIB || CPK != 1, "nine steps per tile: odd");
  using T = bf16raw;
  constexpr int BM = 128, BN = 128, WN = 2, NF = 4, CPL = 16;
  constexpr int PSTR = 32 * 128;
  // (run-time values although CPK fixes them: a DMA's scalar offset must be an SGPR or an inline constant, and sums of run-time
  // scalars stay in SGPRs; the few offsets that ARE literals go through VDQN_SCONST)
  const int tap_k = p.ci * 2;          // = CPK * 128: byte distance between the weight K offsets of consecutive taps of one chunk
  const int b_row32 = 32 * 9 * tap_k;  // 32 weight rows of the 3x3 ([co][3][3][ci] bf16)
  const int b2_row32 = 32 * tap_k;     // ... of the 1x1 ([co][ci])
  const int pixB = tap_k;              // bytes per input pixel (pix_stride == ci)
#define VDQN_SCONST(V) ({ int r_; asm volatile("s_mov_b32 %0, %1" : "=s"(r_) : "n"(V)); r_; })
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tiles of this workgroup: win9u_kernel's XCD-contiguous walk
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t tq = total_tiles >> 3, tr = total_tiles & 7u;
  const uint32_t x_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const uint32_t x_count = tq + (xcd < tr ? 1u : 0u);
  const uint32_t x_blocks = (gridDim.x >> 3) + (xcd < (gridDim.x & 7u) ? 1u : 0u);
  uint32_t lt = blockIdx.x >> 3;
  if (lt >= x_count) return;
  int tile_n = (int)((x_first + lt) % (uint32_t)tiles_n), tile_m = (int)((x_first + lt) / (uint32_t)tiles_n);
  int n0 = tile_n * BN, m0 = tile_m * BM;
  const int Wo = p.wo, rows_total = p.M;
  const int lrow = tid >> 3;
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int lchunk_b = (tid & 7) ^ ((((lrow / CPL) & 1) << 2) | (lrow & 3));

  const unsigned long long a_ptr = (unsigned long long)p.in, b_ptr = (unsigned long long)p.wt, b2_ptr = (unsigned long long)(SIB ? p.wt2 : p.wt);
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};
  const i32x4 rs_b2 = {__builtin_amdgcn_readfirstlane((int)(unsigned)b2_ptr), __builtin_amdgcn_readfirstlane((int)((b2_ptr >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane(SIB ? p.wt2_bytes : p.wt_bytes), 0x00020000};

  // window rows staged by this thread for the tile at m_base (win9s_kernel's a_row): plane pixel q = m_base - Wo - 1 + lrow + 32 i
  const int need = BM + Wo + 1;
  auto plane_rows = [&](int m_base, uint32_t (&ar)[kS_WPass]) {
#pragma unroll
    for (int i = 0; i < kS_WPass; ++i) {
      const int j = lrow + 32 * i;
      const int q = m_base - Wo - 1 + j;
      const bool ok = j < need && (unsigned)q < (unsigned)rows_total;
      const uint32_t qq = ok ? (uint32_t)q : 0u;
      const uint32_t img = fastdiv(qq, d_howo), rem = qq - img * d_howo.div;
      const uint32_t y = fastdiv(rem, d_wo), x = rem - y * d_wo.div;
      ar[i] = ok ? ((img * (uint32_t)p.hi + 2u * y) * (uint32_t)p.wi + 2u * x) * (uint32_t)pixB + (uint32_t)(lchunk_a * 16) : kOob;
    }
  };
  uint32_t a_row[kS_WPass], a_row_n[kS_WPass];
  plane_rows(m0, a_row);
  uint32_t b_off0 = (uint32_t)(n0 + lrow) * (uint32_t)(9 * tap_k) + (uint32_t)(lchunk_b * 16);
  uint32_t b2_off0 = (uint32_t)(n0 + lrow) * (uint32_t)tap_k + (uint32_t)(lchunk_b * 16);

  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * (8 * 128);
  const int so_p11 = (p.wi + 1) * pixB, so_p10 = p.wi * pixB, so_p01 = pixB;  // scalar offsets of the parity planes (P00: 0)

  // window -> window buffer WBUF: five pieces, per-lane rows AR_ (this tile's or the next tile's), scalar offset = plane + chunk
#define VDQN_P_ISSUE_AW(WBUF, AR_, SO_)                                                                            \
  {                                                                                                                \
    const uint32_t la_ = lds_wave + (uint32_t)(kS_WinBase + (WBUF)*kS_WinStride);                                  \
    const int so_ = (SO_);                                                                                         \
    asm volatile(                                                                                                  \
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %6, %7 offen lds\n\t"                             \
        "s_add_u32 m0, %5, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %6, %7 offen lds\n\t"                         \
        "s_add_u32 m0, %5, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %7 offen lds\n\t"                         \
        "s_add_u32 m0, %5, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %7 offen lds\n\t"                        \
        "s_add_u32 m0, %5, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %7 offen lds"                             \
        ::"v"(AR_[0]), "v"(AR_[1]), "v"(AR_[2]), "v"(AR_[3]), "v"(AR_[4]), "s"(la_), "s"(rs_a), "s"(so_),          \
          "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR), "n"(4 * PSTR)                                                   \
        : "memory", "scc");                                                                                        \
  }
  // weight tile -> weight buffer BUF: four pieces (rows lrow + 32 i of the column tile at per-lane offset VOFF_ of descriptor RS_,
  // K offset SO0_, RSTR_ bytes per 32 rows)
#define VDQN_P_ISSUE_B(BUF, VOFF_, RS_, SO0_, RSTR_)                                                               \
  {                                                                                                                \
    const uint32_t lb_ = lds_wave + (uint32_t)((BUF)*kS_WtTile);                                                   \
    const i32x4 rsb_ = (RS_);                                                                                      \
    const int so0_ = (SO0_), so1_ = so0_ + (RSTR_), so2_ = so1_ + (RSTR_), so3_ = so2_ + (RSTR_);                  \
    asm volatile(                                                                                                  \
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\t"                             \
        "s_add_u32 m0, %1, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %4 offen lds\n\t"                         \
        "s_add_u32 m0, %1, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %5 offen lds\n\t"                         \
        "s_add_u32 m0, %1, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %6 offen lds"                              \
        ::"v"(VOFF_), "s"(lb_), "s"(rsb_), "s"(so0_), "s"(so1_), "s"(so2_), "s"(so3_), "n"(PSTR), "n"(2 * PSTR), "n"(3 * PSTR) \
        : "memory", "scc");                                                                                        \
  }

  f32x4 acc[4][NF];
  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  auto edge_bits = [&](int m_base) {  // 2 bits per fragment f: 1 top row (dy = -1 leaves the image), 2 left column (dx = -1)
    uint32_t eb = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint32_t m = (uint32_t)(m_base + wr * 64 + f * 16 + i16);
      const uint32_t rem = m - fastdiv(m, d_howo) * d_howo.div;
      const uint32_t oh = fastdiv(rem, d_wo), ow = rem - oh * d_wo.div;
      eb |= ((oh == 0 ? 1u : 0u) | (ow == 0 ? 2u : 0u)) << (2 * f);
    }
    return eb;
  };
  uint32_t edge8 = edge_bits(m0);
  uint32_t ab[4];  // [2 * (dy + 1) + (dx + 1)]
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int joff = Wo + 1 - ((c & 2) ? 0 : Wo) - ((c & 1) ? 0 : 1);
    const int row = wr * 64 + i16 + joff;
    ab[c] = (uint32_t)(row * 128 + ((g ^ ((i16 + joff) & 7)) << 4));
  }
  const uint32_t bb0 = (uint32_t)((wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128 + ((g ^ (i16 & 7)) << 4));
  const uint32_t bb1 = (uint32_t)((wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128 + (((g + 4) ^ (i16 & 7)) << 4));

  u32x4 fa[2][2][4], fb[2][2][NF];  // [register set][K half][fragment]
  // fragments: window buffer WBUF_, shift class (DY_, DX_), weight buffer BBUF_ -> register set SET
#define VDQN_P_LOAD_FRAGS(SET, WBUF_, DY_, DX_, BBUF_)                                                             \
  {                                                                                                                \
    constexpr uint32_t tb_ = ((DY_) < 0 ? 1u : 0u) | ((DX_) < 0 ? 2u : 0u);                                        \
    const unsigned char* wb_ = smem + kS_WinBase + (WBUF_)*kS_WinStride;                                           \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                             \
      uint32_t a0_ = ab[2 * ((DY_) + 1) + ((DX_) + 1)];                                                            \
      if constexpr (tb_ != 0u) {                                                                                   \
        const bool z_ = (edge8 & (tb_ << (2 * f_))) != 0u;                                                         \
        a0_ = z_ ? ((a0_ & 255u) | (uint32_t)((kS_WinRows - 2) * 128 - f_ * 16 * 128)) : a0_;                      \
      }                                                                                                            \
      const uint32_t a1_ = a0_ ^ 64u;                                                                              \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a0_);                                 \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 16 * 128 + a1_);                                 \
    }                                                                                                              \
    const unsigned char* bt_ = smem + (BBUF_)*kS_WtTile;                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                            \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + bb0);                                  \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 4 * 128 + bb1);                                  \
    }                                                                                                              \
  }
#define VDQN_P_LOAD_STEP(SET, S_, BBUF_) VDQN_P_LOAD_FRAGS(SET, s2_step(S_).wbuf, s2_step(S_).dy, s2_step(S_).dx, BBUF_)
#define VDQN_P_MFMA_ALL(SET)                                                                                       \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)               \
      _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                          \
    acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[SET][h_][j_]),             \
                                                          __builtin_bit_cast(bf16x8, fa[SET][h_][f_]), acc[f_][j_], 0, 0, 0); \
  }
#ifdef VDQN_STAMP
  // diagnostic build only (tools/stamp_s2.py): s_memtime deltas summed per workgroup by wave 0 — the wait at the top of a K-step
  // (first step behind an epilogue / second / any other), the barrier, the step's DMA issue, fragment reads + MFMAs, the phase
  // boundary's wait + barrier, the two epilogues, the tile's top
  unsigned long long st_wait = 0, st_wait0 = 0, st_wait1 = 0, st_bar = 0, st_issue = 0, st_comp = 0, st_bnd = 0, st_epi1 = 0, st_epi2 = 0, st_top = 0, st_t = 0;
  unsigned long long st_tiles = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long st_rt_begin = __builtin_amdgcn_s_memrealtime();
#define VDQN_PST(ACC)                                                    \
  {                                                                      \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();          \
    ACC += n_ - st_t;                                                    \
    st_t = n_;                                                           \
  }
#else
#define VDQN_PST(ACC)
#endif
  // top of a K-step whose fragments are in register set CUR_: own DMA landed, own reads complete, everyone past the buffers
#define VDQN_P_TOP(CUR_, WACC_)                                                                                    \
  VDQN_PST(st_comp)                                                                                                \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                      \
  asm volatile("" : "+v"(edge8));                                                                                  \
  VDQN_PST(WACC_)                                                                                                  \
  asm volatile("" : "+v"(fa[CUR_][0][0]), "+v"(fa[CUR_][0][1]), "+v"(fa[CUR_][0][2]), "+v"(fa[CUR_][0][3]),        \
                    "+v"(fa[CUR_][1][0]), "+v"(fa[CUR_][1][1]), "+v"(fa[CUR_][1][2]), "+v"(fa[CUR_][1][3]));       \
  _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[CUR_][0][j_]), "+v"(fb[CUR_][1][j_])); \
  __builtin_amdgcn_s_barrier();                                                                                    \
  VDQN_PST(st_bar)
#define VDQN_P_BODY(CUR_, LOAD_)                                                                                   \
  VDQN_PST(st_issue)                                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  LOAD_                                                                                                            \
  VDQN_P_MFMA_ALL(CUR_)                                                                                            \
  VDQN_INTERLEAVE(8 + 2 * NF)                                                                                      \
  __builtin_amdgcn_sched_barrier(0);

#ifdef VDQN_STAMP
#define VDQN_P_TOP_U(U, C0_, CUR_)                                       \
  if constexpr ((U) == 0 && (C0_) == 0) { VDQN_P_TOP(CUR_, st_wait0) }   \
  else if constexpr ((U) == 1 && (C0_) == 0) { VDQN_P_TOP(CUR_, st_wait1) } \
  else { VDQN_P_TOP(CUR_, st_wait) }
#define VDQN_P_TOP_D(D, CUR_)                                            \
  if constexpr ((D) == 0) { VDQN_P_TOP(CUR_, st_wait0) }                 \
  else if constexpr ((D) == 1) { VDQN_P_TOP(CUR_, st_wait1) }            \
  else { VDQN_P_TOP(CUR_, st_wait) }
#else
#define VDQN_P_TOP_U(U, C0_, CUR_) VDQN_P_TOP(CUR_, st_wait)
#define VDQN_P_TOP_D(D, CUR_) VDQN_P_TOP(CUR_, st_wait)
#endif
  // 3x3 K-step U of a block that starts at chunk C0_ (PAIR_: 18 steps over chunks C0_, C0_ + 1; else the single 9-step block of the
  // one-chunk layer).  The two steps whose staging target lies behind the block's end (PAIR_: U = 16, 17; single: U = 7, 8) stage
  // the next chunk pair (LAST_ = false), or behind the tile's last chunk (LAST_, compile time: the pair body is instantiated once per
  // iteration) the downsample's first steps (SIB) / the next tile's first steps.
#define VDQN_P_USTEP(U, C0_, PAIR_, LAST_)                                                                         \
  {                                                                                                                \
    constexpr int cur_ = (U)&1, nxt_ = cur_ ^ 1;                                                                   \
    constexpr int s2_ = ((U) + 2) % 9, c2_ = ((U) + 2) / 9;                                                        \
    constexpr int s1_ = ((U) + 1) % 9;                                                                             \
    constexpr int wi_ = s2_window_issued_at((U) % 9);                                                              \
    constexpr bool behind_ = (U) + 2 >= ((PAIR_) ? 18 : 9);                                                        \
    constexpr int e_ = (U) + 2 - ((PAIR_) ? 18 : 9); /* behind_: 0 or 1 = which step behind the end */             \
    VDQN_P_TOP_U(U, C0_, cur_)                                                                                             \
    if constexpr (!behind_) {                                                                                      \
      VDQN_P_ISSUE_B(cur_, b_off0, rs_b, s2_step(s2_).tap * tap_k + ((C0_) + c2_) * 128, b_row32)                  \
      if constexpr (wi_ >= 0) {                                                                                    \
        constexpr S2Step w_ = s2_step(wi_ >= 0 ? wi_ % 9 : 0);                                                     \
        VDQN_P_ISSUE_AW(w_.wbuf, a_row, (w_.a ? so_p10 : 0) + (w_.b ? so_p01 : 0) + ((C0_) + ((U) / 9) + wi_ / 9) * 128) \
      }                                                                                                            \
    } else if constexpr (!(LAST_)) { /* chunk C0_ + 2 of this tile (pair bodies only) */                           \
      VDQN_P_ISSUE_B(cur_, b_off0, rs_b, s2_step(e_).tap * tap_k + ((C0_) + 2) * 128, b_row32)                     \
      if constexpr (e_ == 0) VDQN_P_ISSUE_AW(0, a_row, so_p11 + ((C0_) + 2) * 128)                                 \
    } else if constexpr (SIB && (PAIR_)) { /* D0 / D1: weight tiles 0 / 1 of wt2; D0's P00 window of chunk 0 -> buffer 0; D1's  \
                                              window is chunk 1's P00: resident in buffer 1 (two chunks) or staged now (four) */ \
      VDQN_P_ISSUE_B(cur_, b2_off0, rs_b2, e_ * 128, b2_row32)                                                     \
      if constexpr (e_ == 0) VDQN_P_ISSUE_AW(0, a_row, 0)                                                          \
      if constexpr (e_ == 1 && CPK == 4) VDQN_P_ISSUE_AW(1, a_row, VDQN_SCONST(128))                               \
    } else if constexpr (SIB) { /* one chunk: step 7 stages D0's weight tile (its window is step 8's), step 8 the next tile's step 0 */ \
      if constexpr (e_ == 0) {                                                                                     \
        VDQN_P_ISSUE_B(cur_, b2_off0, rs_b2, 0, b2_row32)                                                          \
      } else {                                                                                                     \
        VDQN_P_ISSUE_B(cur_, b_t, rs_b, s2_step(0).tap * tap_k, b_row32)                                           \
        VDQN_P_ISSUE_AW(0, a_row_n, so_p11)                                                                        \
      }                                                                                                            \
    } else { /* no sibling: the next tile's steps 0 / 1 */                                                         \
      VDQN_P_ISSUE_B(cur_, b_t, rs_b, s2_step(e_).tap * tap_k, b_row32)                                            \
      if constexpr (e_ == 0) VDQN_P_ISSUE_AW(0, a_row_n, so_p11)                                                   \
    }                                                                                                              \
    VDQN_P_BODY(cur_, VDQN_P_LOAD_STEP(nxt_, s1_, nxt_))                                                           \
  }
#define VDQN_P_PAIR(C0_, LAST_)                                                                                                \
  VDQN_P_USTEP(0, C0_, true, LAST_) VDQN_P_USTEP(1, C0_, true, LAST_) VDQN_P_USTEP(2, C0_, true, LAST_) VDQN_P_USTEP(3, C0_, true, LAST_)   \
  VDQN_P_USTEP(4, C0_, true, LAST_) VDQN_P_USTEP(5, C0_, true, LAST_) VDQN_P_USTEP(6, C0_, true, LAST_) VDQN_P_USTEP(7, C0_, true, LAST_)   \
  VDQN_P_USTEP(8, C0_, true, LAST_) VDQN_P_USTEP(9, C0_, true, LAST_) VDQN_P_USTEP(10, C0_, true, LAST_) VDQN_P_USTEP(11, C0_, true, LAST_) \
  VDQN_P_USTEP(12, C0_, true, LAST_) VDQN_P_USTEP(13, C0_, true, LAST_) VDQN_P_USTEP(14, C0_, true, LAST_) VDQN_P_USTEP(15, C0_, true, LAST_) \
  VDQN_P_USTEP(16, C0_, true, LAST_) VDQN_P_USTEP(17, C0_, true, LAST_)
  // downsample step D (global step 9 CPK + D): P00 window of chunk D in window buffer WB(D), weight tile D of wt2
#define VDQN_P_DSTEP(D)                                                                                            \
  {                                                                                                                \
    constexpr int cur_ = (9 * CPK + (D)) & 1, nxt_ = cur_ ^ 1;                                                     \
    constexpr int d2_ = (D) + 2;                                                                                   \
    VDQN_P_TOP_D(D, cur_)                                                                                             \
    if constexpr (d2_ < CPK) {                                                                                     \
      VDQN_P_ISSUE_B(cur_, b2_off0, rs_b2, VDQN_SCONST(d2_ * 128), b2_row32)                                       \
      VDQN_P_ISSUE_AW(d2_ & 1, a_row, VDQN_SCONST(d2_ * 128))                                                      \
    } else { /* the next tile's K-step d2_ - CPK (0 or 1) */                                                       \
      VDQN_P_ISSUE_B(cur_, b_t, rs_b, s2_step(d2_ - CPK).tap * tap_k, b_row32)                                     \
      if constexpr (d2_ - CPK == 0) VDQN_P_ISSUE_AW(0, a_row_n, so_p11)                                            \
    }                                                                                                              \
    VDQN_P_BODY(cur_, VDQN_P_LOAD_FRAGS(nxt_, CPK == 1 ? 1 : (((D) + 1) & 1), 0, 0, nxt_))                         \
  }

  // ---- lean epilogue (round 5).  The shared igemm_epilogue cost 8.1 k cycles per call here (stamps, profiles/r05b_stamp_s2.txt:
  // 38 % of a layer2.0 tile in two calls): sixteen bias loads from global memory with their latency exposed, 64-bit address
  // arithmetic, generic branches.  This kernel's calls have bias + ReLU only and whole 16-byte vectors, so: the layer's biases are
  // staged in LDS once per workgroup, the stores are buffer stores with 32-bit offsets (rows behind M get an out-of-range offset and
  // are dropped).  Same arithmetic, same bits. ----
  float* sBias = reinterpret_cast<float*>(smem + kS_Smem);  // [2][kSP_BiasCols]: 3x3, sibling
  for (int i = tid; i < p.co && i < kSP_BiasCols; i += 256) {
    sBias[i] = p.bias ? p.bias[i] : 0.f;
    if constexpr (SIB) sBias[kSP_BiasCols + i] = p.bias2 ? p.bias2[i] : 0.f;
  }
  const bool lean = !p.no_lean && p.vec_ok && !p.resid && !p.out_f32 && !p.colsum_part && !p.mask && p.out && p.co <= kSP_BiasCols &&
                    (long long)p.M * p.ldo * 2 < 0x7fffffffLL && (!SIB || (long long)p.M * p.ldo2 * 2 < 0x7fffffffLL);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, lean ? (int)((long long)p.M * p.ldo * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out2 =
      __builtin_amdgcn_make_buffer_rsrc(SIB ? p.out2 : p.out, 0, (lean && SIB) ? (int)((long long)p.M * p.ldo2 * 2) : 0, 0x00020000);
  auto lean_epilogue = [&](const __amdgpu_buffer_rsrc_t rsrc, const int ldo, const int relu, const float* sb) {
    const int ncol = n0 + wc * (BN / WN) + g * CPL;
    float bv[CPL];
#pragma unroll
    for (int e4 = 0; e4 < CPL / 4; ++e4) *reinterpret_cast<float4*>(bv + 4 * e4) = *reinterpret_cast<const float4*>(sb + ncol + 4 * e4);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int m = m0 + wr * 64 + f * 16 + i16;
      const uint32_t off = m < rows_total ? (uint32_t)(m * ldo + ncol) * 2u : kOob;
      bf16raw ov[CPL];
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[f][j][r] + bv[j * 4 + r];
          if (relu) v = fmaxf(v, 0.f);
          ov[j * 4 + r] = f32_to_bf16(v);
        }
      __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const u32x4*>(ov)[0], rsrc, (int)off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const u32x4*>(ov)[1], rsrc, (int)off + 16, 0, 0);
    }
  };

  IgemmParams q = p;  // the sibling's epilogue: its bias, output and ReLU flag; no residual, mask, f32 copy or column sums
  if constexpr (SIB) {
    q.bias = p.bias2; q.out = p.out2; q.relu = p.relu2; q.co = p.co2; q.ldo = p.ldo2;
    q.resid = nullptr; q.mask = nullptr; q.out_f32 = nullptr; q.colsum_part = nullptr;
  }

  // prologue of the workgroup's first tile: K-steps 0 and 1 of chunk 0 (P11's window, weight tiles of taps 0 and 2)
  VDQN_P_ISSUE_B(0, b_off0, rs_b, s2_step(0).tap * tap_k, b_row32)
  VDQN_P_ISSUE_AW(0, a_row, so_p11)
  VDQN_P_ISSUE_B(1, b_off0, rs_b, s2_step(1).tap * tap_k, b_row32)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef VDQN_STAMP
  st_t = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {  // tiles of this workgroup
#ifdef VDQN_STAMP
    ++st_tiles;
#endif
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    VDQN_P_LOAD_STEP(0, 0, 0)  // the fragments of step 0
    // the next tile of this workgroup (if any; else this one again: what is staged for it is never read)
    const uint32_t lt_nx = lt + x_blocks;
    const bool has_nx = lt_nx < x_count;
    const int tn_nx = has_nx ? (int)((x_first + lt_nx) % (uint32_t)tiles_n) : tile_n;
    const int tm_nx = has_nx ? (int)((x_first + lt_nx) / (uint32_t)tiles_n) : tile_m;
    plane_rows(tm_nx * BM, a_row_n);
    const uint32_t b_t = (uint32_t)(tn_nx * BN + lrow) * (uint32_t)(9 * tap_k) + (uint32_t)(lchunk_b * 16);
    VDQN_PST(st_top)
    if constexpr (CPK == 1) {
      // one chunk + the downsample: steps 0..8, then D0 on step 8's P00 window
      VDQN_P_USTEP(0, 0, false, true) VDQN_P_USTEP(1, 0, false, true) VDQN_P_USTEP(2, 0, false, true) VDQN_P_USTEP(3, 0, false, true)
      VDQN_P_USTEP(4, 0, false, true) VDQN_P_USTEP(5, 0, false, true) VDQN_P_USTEP(6, 0, false, true) VDQN_P_USTEP(7, 0, false, true)
      VDQN_P_USTEP(8, 0, false, true)
    } else if constexpr (CPK == 2) {
      VDQN_P_PAIR(0, true)
    } else {
      VDQN_P_PAIR(0, false)
      VDQN_P_PAIR(2, true)
    }
    VDQN_PST(st_comp)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // what the last two steps staged has landed; the stale fragment reads too
    __builtin_amdgcn_s_barrier();
    VDQN_PST(st_bnd)
    // (the epilogue's LDS scratch serves column sums only, which the dispatch keeps off this kernel)
    if (lean) lean_epilogue(r_out, p.ldo, p.relu, sBias);
    else igemm_epilogue<T, BM, BN, 0, WN>(p, acc, smem + kS_WinBase, m0, n0, tile_m, rows_total, p.howo, Wo, 0, 0, p.bias);
    VDQN_PST(st_epi1)
    if constexpr (SIB) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // D0's fragments: register set (9 CPK) & 1; one chunk: the P00 window of step 8 (buffer 1), else buffer 0; weight buffer = set
      VDQN_P_LOAD_FRAGS((9 * CPK) & 1, CPK == 1 ? 1 : 0, 0, 0, (9 * CPK) & 1)
      VDQN_P_DSTEP(0)
      if constexpr (CPK >= 2) VDQN_P_DSTEP(1)
      if constexpr (CPK >= 4) { VDQN_P_DSTEP(2) VDQN_P_DSTEP(3) }
      VDQN_PST(st_comp)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      VDQN_PST(st_bnd)
      if (lean) lean_epilogue(r_out2, p.ldo2, p.relu2, sBias + kSP_BiasCols);
      else igemm_epilogue<T, BM, BN, 0, WN>(q, acc, smem + kS_WinBase, m0, n0, tile_m, rows_total, p.howo, Wo, 0, 0, q.bias);
      VDQN_PST(st_epi2)
    }
    if (!has_nx) break;
    lt = lt_nx;
    tile_n = tn_nx; tile_m = tm_nx;
    n0 = tile_n * BN; m0 = tile_m * BM;
#pragma unroll
    for (int i = 0; i < kS_WPass; ++i) a_row[i] = a_row_n[i];
    b_off0 = b_t;
    b2_off0 = (uint32_t)(n0 + lrow) * (uint32_t)tap_k + (uint32_t)(lchunk_b * 16);
    edge8 = edge_bits(m0);
  }
#undef VDQN_P_DSTEP
#undef VDQN_SCONST
#undef VDQN_P_PAIR
#undef VDQN_P_USTEP
#undef VDQN_P_BODY
#undef VDQN_P_TOP
#undef VDQN_P_MFMA_ALL
#undef VDQN_P_LOAD_STEP
#undef VDQN_P_LOAD_FRAGS
#undef VDQN_P_ISSUE_B
#undef VDQN_P_ISSUE_AW
#ifdef VDQN_STAMP
  if (stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last epilogue's stores have left
    unsigned long long* o = reinterpret_cast<unsigned long long*>(stamps) + (size_t)blockIdx.x * 16;
    o[0] = st_begin; o[1] = __builtin_amdgcn_s_memtime(); o[2] = st_tiles; o[3] = st_wait; o[4] = st_bar; o[5] = st_issue; o[6] = st_comp;
    o[7] = st_epi1; o[8] = st_epi2; o[9] = st_bnd; o[10] = st_rt_begin; o[11] = __builtin_amdgcn_s_memrealtime();
    o[12] = st_top; o[13] = st_wait0; o[14] = st_wait1;
  }
#endif
#undef VDQN_PST
#undef VDQN_P_TOP_U
#undef VDQN_P_TOP_D
}

}  // namespace

#ifdef VDQN_STAMP
extern void* g_stamp_buffer;
#endif

// entry used by vdqn_conv2d (igemm.hip) for forward 3x3 / stride 2 / pad 1 over an even-sized input, bf16, 128-column tiles;
// p.wt2 != nullptr: the sibling 1x1 / stride-2 convolution (p.wt2 / bias2 / out2 / relu2, co2 == co) in the same launch
template <int CPK, bool SIB>
static void launch_win9sp(const IgemmParams& p, unsigned tiles, int tiles_n, hipStream_t stream) {
  // VDQN_S2WIN_PERSIST: 1 (default) = persistent workgroups when the launch has more tiles than the chip holds at once (two
  // workgroups per CU), 2 = when it has more than two rounds of them, 0 = one workgroup per tile (same kernel, no tile walk)
  static const int persist = [] { const char* e = getenv("VDQN_S2WIN_PERSIST"); return e ? atoi(e) : 1; }();
  const unsigned resident = 2u * (unsigned)vdqn_num_cus();
  const unsigned grid = ((persist == 1 && tiles > resident) || (persist >= 2 && tiles > 2 * resident)) ? resident : tiles;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9sp_kernel<CPK, SIB>), (size_t)kSP_Smem);
  void* stamps = nullptr;
#ifdef VDQN_STAMP
  stamps = g_stamp_buffer;
#endif
  hipLaunchKernelGGL((win9sp_kernel<CPK, SIB>), dim3(grid), dim3(256), kSP_Smem, stream, p, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo), tiles, tiles_n,
                     stamps);
}

// which (channel chunks, sibling) combinations the plane-window kernels take (igemm.hip asks before routing a call here)
int vdqn_win9s_supports(int cpk, int has_sib) { return has_sib ? (cpk == 1 || cpk == 2 || cpk == 4) : cpk >= 1; }

int vdqn_launch_win9s(const void* pv, hipStream_t stream) {
  IgemmParams p = *reinterpret_cast<const IgemmParams*>(pv);
  const bool sib = p.wt2 != nullptr;
  const int cpk = p.ci / 64;
  const int tiles_n = sib ? p.tiles_n1 : p.tiles_n;  // (with a sibling igemm.hip counts its column tiles behind the 3x3's)
  const unsigned tiles = (unsigned)(((p.M + 127) / 128) * tiles_n);
  p.wt_bytes = (int)((long long)tiles_n * 128 * p.ktot * 2);
  static const int persist_kernel = [] { const char* e = getenv("VDQN_S2WIN_PERSIST"); return e ? atoi(e) : 1; }();
  vdqn_prof_begin("igemm_s2win<bf16,128,fwd>", 2.0 * p.M * p.co * p.ktot + (sib ? 2.0 * p.M * p.co2 * p.ci : 0.0),
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr)) +
                         (sib ? (double)p.co2 * p.ci + (double)p.M * p.co2 : 0.0)), stream);
  const bool templ = cpk == 2 || cpk == 4 || (cpk == 1 && sib);
  if (templ && (sib || persist_kernel != -1)) {
    if (cpk == 1) launch_win9sp<1, true>(p, tiles, tiles_n, stream);
    else if (cpk == 2) { if (sib) launch_win9sp<2, true>(p, tiles, tiles_n, stream); else launch_win9sp<2, false>(p, tiles, tiles_n, stream); }
    else { if (sib) launch_win9sp<4, true>(p, tiles, tiles_n, stream); else launch_win9sp<4, false>(p, tiles, tiles_n, stream); }
  } else {
    // other chunk counts, and the one-chunk layer without a sibling (9 steps per tile): one workgroup per tile
    // (VDQN_S2WIN_PERSIST=-1 keeps this kernel for every call without a sibling: the round-4 behaviour)
    vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9s_kernel), (size_t)kS_Smem);
    hipLaunchKernelGGL(win9s_kernel, dim3(tiles), dim3(256), kS_Smem, stream, p, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo));
  }
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
