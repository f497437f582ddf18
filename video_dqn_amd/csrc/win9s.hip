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

}  // namespace

// entry used by vdqn_conv2d (igemm.hip) for forward 3x3 / stride 2 / pad 1 over an even-sized input, bf16, 128-column tiles
int vdqn_launch_win9s(const void* pv, hipStream_t stream) {
  const IgemmParams& p = *reinterpret_cast<const IgemmParams*>(pv);
  const unsigned tiles = (unsigned)(((p.M + 127) / 128) * p.tiles_n);
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9s_kernel), (size_t)kS_Smem);
  vdqn_prof_begin("igemm_s2win<bf16,128,fwd>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr))), stream);
  hipLaunchKernelGGL(win9s_kernel, dim3(tiles), dim3(256), kS_Smem, stream, p, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo));
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
