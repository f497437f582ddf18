// Nine-tap window kernel on 32 x 32 x 16 MFMAs (bf16; MODE 0 forward, MODE 1 data gradient): the same tiling, staging, K order and
// persistent tile walk as win9m_kernel (win9.hip) — 128 x 128 tiles, 4 waves of 64 x 64, one staged window per 64-channel chunk
// for all nine taps, two LDS buffers, two register sets of fragments, one barrier per K-step, two workgroups per CU — with the
// wave tile computed by 2 x 2 v_mfma_f32_32x32x16_bf16 tiles instead of 4 x 4 v_mfma_f32_16x16x32_bf16.  A K-step is then 16
// MFMA issues of 32 cycles instead of 32 of 16: an MFMA holds the SIMD's vector issue port for 8 of its cycles whatever its shape
// (MI355X_MICROARCH.md, cycle constants), so the fragment reads (the same 16 ds_read_b128), the zero selects and the LDS-DMA
// issue of a step have 24 instead of 8 free issue cycles per MFMA to hide in.  Against that stands the DVFS note of the same
// guide (bare 32x32x16 loops hold a lower clock than 16x16x32 loops on random data): which shape wins is measured, not assumed
// (VDQN_WIN9_MFMA32 selects this kernel; profiles/r03*_mfma32_ab.txt).
//
// Fragment layout (lane = 32 hq + i32): the activation operand (MFMA columns = pixels) of pixel block pb and K sub-step s is the
// 16 bytes of window row (64 wr + 32 pb + i32 + tap shift), K chunk hq + 2 s; the weight operand (MFMA rows = channels) of channel
// block cb is weight row 64 wc + 32 cb + n(i32), n(r) = 16 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3) — with that permutation the 16
// accumulator registers of a lane are 16 CONSECUTIVE output channels (32 cb + 16 hq ..) of pixel 32 pb + i32, so the epilogue
// goes straight from the accumulators to 16-byte global accesses, as in the other kernels.  LDS swizzles (source side of the
// LDS-DMA, the same XOR on the read): activations chunk ^ ((row & 7) ^ ((row >> 3) & 1)), weights chunk ^ ((row >> 1) & 7) — both
// conflict-free for the ds_read_b128 lane groups at every tap shift (bank model: tools/bank_model.py).
//
// Serves the same reference call sites as win9.hip: torchvision BasicBlock conv1 / conv2 reached from
// archs/HabitatDQNMultiAction.py:30,49-51 and their data gradient (train_q_network.py:226).
#include <stdlib.h>

#include "igemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// MFMA / VALU / LDS interleave of a K-step: each of the 16 MFMAs is followed by two VALU and one LDS read of the next step
#define VDQN_INTERLEAVE32()                                 \
  _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
  }

// Epilogue straight from the 32 x 32 accumulator tiles (the arithmetic of igemm_epilogue): lane (i32, hq) of wave (wr, wc) owns the
// 16 consecutive channels n0 + 64 wc + 32 cb + 16 hq .. of the two pixels m0 + 64 wr + 32 pb + i32 — bias, residual, ReLU, ReLU mask,
// bf16 and / or f32 stores as 16-byte vectors, per-tile column sums of the stored values.
template <int BM>
__device__ __forceinline__ void win9m_epilogue(const IgemmParams& p, f32x16 (&acc)[2][2], unsigned char* smem, int m0, int n0, int tile_m, int rows_total,
                                               const float* __restrict__ bias) {
  using T = bf16raw;
  constexpr int BN = 128, WN = 2;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int i32 = lane & 31, hq = lane >> 5;
  T* __restrict__ out = (T*)p.out;
  const T* __restrict__ resid = (const T*)p.resid;
  const T* __restrict__ mask = (const T*)p.mask;
  float cs[2][16];  // per-lane column sums of the stored values, per channel block
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[j][e] = 0.f;
  // (the launcher only takes layers whose channel count is a multiple of the 128-column tile and whose rows are 16-byte aligned:
  // every lane's 16 channels exist and vector accesses are legal)
  size_t o[2];
  bool okr[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int m = m0 + wr * 64 + f * 32 + i32;
    okr[f] = m < rows_total;
    o[f] = (size_t)m * p.ldo + n0 + wc * (BN / WN) + hq * 16;
  }
  uint4 rv[2][2][2], mv[2][2][2];
  if (resid) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) rv[f][j][q] = okr[f] ? reinterpret_cast<const uint4*>(resid + o[f] + j * 32)[q] : make_uint4(0, 0, 0, 0);
  }
  if (mask) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) mv[f][j][q] = okr[f] ? reinterpret_cast<const uint4*>(mask + o[f] + j * 32)[q] : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = bias ? bias[n0 + wc * (BN / WN) + j * 32 + hq * 16 + e] : 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[f][j][e] + bv[e];
      if (resid) {
        const T* pr = reinterpret_cast<const T*>(rv[f][j]);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += to_f32<T>(pr[e]);
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (mask) {
        const T* pm = reinterpret_cast<const T*>(mv[f][j]);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (to_f32<T>(pm[e]) > 0.f) ? v[e] : 0.f;
      }
      if (okr[f]) {
        if (out) {
          T ov[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            ov[e] = from_f32<T>(v[e]);
            cs[j][e] += to_f32<T>(ov[e]);
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) reinterpret_cast<uint4*>(out + o[f] + j * 32)[q] = reinterpret_cast<const uint4*>(ov)[q];
        }
        if (p.out_f32) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(p.out_f32 + o[f] + j * 32 + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
      }
    }
  }
  if (p.colsum_part) {  // uniform branch: partial column sums of this tile -> colsum_part[tile_m][ldo]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {  // the 32 pixel lanes of every channel
        float t = cs[j][e];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        cs[j][e] = t;
      }
    __syncthreads();  // every wave is past its last fragment read: LDS can be reused
    float* sR = reinterpret_cast<float*>(smem);
    if (i32 == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sR[wr * BN + wc * (BN / WN) + j * 32 + hq * 16 + e] = cs[j][e];
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.co) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < BM / 64; ++r) t += sR[r * BN + tid];
      if constexpr (BM == 256) {  // consumers sum ceil(M/128) entries: this tile covers two of them
        p.colsum_part[(size_t)(2 * tile_m) * p.ldo + n0 + tid] = t;
        if ((2 * tile_m + 1) * 128 < p.M) p.colsum_part[(size_t)(2 * tile_m + 1) * p.ldo + n0 + tid] = 0.f;
      } else {
        p.colsum_part[(size_t)tile_m * p.ldo + n0 + tid] = t;
      }
    }
  }
}

constexpr int kM_WtTile = 128 * 128;       // one staged weight tile
constexpr int kM_WinBase = 2 * kM_WtTile;  // LDS: [2 weight tiles][2 windows]

// BM = 128: 4 waves, two workgroups per CU.  BM = 256: 8 waves (4 x 2 of 64 x 64), one workgroup per CU — the two co-resident
// 128-row tiles of a CU made one, so that the weight tile (16 of the 18.7 KB a 128-row tile stages per K-step) is staged once for
// both halves: 21.4 KB per K-step and CU instead of 37.4 KB.
template <int BM>
struct Win9mGeom {
  static constexpr int NT = 2 * BM;                     // threads
  static constexpr int RPP = NT / 8;                    // rows one staging pass of the workgroup covers (8 lanes x 16 B per row)
  static constexpr int PSTR = RPP * 128;                // LDS distance between a thread's consecutive DMA pieces
  static constexpr int WinRows = BM == 128 ? 192 : 320;  // >= BM + 2 * 28 + 3, a multiple of RPP
  static constexpr int WinStride = WinRows * 128;       // bytes between the two window buffers
  static constexpr int WPass = WinRows / RPP;           // 6 / 5 staging passes per window
  static constexpr int BPass = 128 / RPP;               // 4 / 2 per weight tile
  static constexpr int Smem = kM_WinBase + 2 * WinStride;
};

template <int MODE, int BM>
__global__ __launch_bounds__(2 * BM, 2) void win9m_kernel(const IgemmParams p, const int wrows, const FastDiv d_wo, const FastDiv d_howo, const uint32_t total_tiles, void* stamps) {
  static_assert(MODE == 0 || MODE == 1, "window kernel: forward or stride-1 data gradient");
  static_assert(BM == 128 || BM == 256, "tile rows");
  using T = bf16raw;  // (VDQN_INTERLEAVE keys on sizeof(T))
  using G = Win9mGeom<BM>;
  constexpr int BN = 128, WN = 2;
  constexpr int NF = BN / (16 * WN);  // 4
  constexpr int CPL = 4 * NF;         // 16
  constexpr int PSTR = G::PSTR;
  constexpr int kM_WinStride = G::WinStride;
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
  if (lt >= x_count) return;
  int tile_n = (int)((x_first + lt) % (uint32_t)p.tiles_n), tile_m = (int)((x_first + lt) / (uint32_t)p.tiles_n);
  int n0 = tile_n * BN, m0 = tile_m * BM;
  const int W = p.wo, H = p.ho, rows_total = p.M;
  const int lrow = tid >> 3;
  // source-side swizzle of the LDS-DMA (rows lrow + RPP i: RPP is a multiple of 32, so the keys are the same for every piece)
  const int lchunk_a = (tid & 7) ^ ((lrow & 7) ^ ((lrow >> 3) & 1));
  const int lchunk_b = (tid & 7) ^ ((lrow >> 1) & 7);

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
  const int m_split = MODE == 0 ? p.m_split : 0x7fffffff;
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
    const uint32_t la_ = lds_wave + (uint32_t)(kM_WinBase + (WBUF)*kM_WinStride);                                   \
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
    const uint32_t lb_ = lds_wave + (uint32_t)((BUF)*kM_WtTile);                                                    \
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

  f32x16 acc[2][2];  // [pixel block pb][channel block cb]
  const int wr = wave / WN, wc = wave % WN;
  const int i32 = lane & 31, hq = lane >> 5;
  // edge bits of this lane's two pixels (of the tile at m_base), 4 bits per pixel block pb: 1 top row, 2 bottom row, 4 left
  // column, 8 right column
  auto edge_bits = [&](int m_base) {
    uint32_t eb = 0;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint32_t m = (uint32_t)(m_base + wr * 64 + f * 32 + i32);
      const uint32_t rem = m - fastdiv(m, d_howo) * d_howo.div;
      const uint32_t oh = fastdiv(rem, d_wo), ow = rem - oh * d_wo.div;
      const uint32_t e = (oh == 0 ? 1u : 0u) | (oh == (uint32_t)H - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == (uint32_t)W - 1 ? 8u : 0u);
      eb |= e << (4 * f);
    }
    return eb;
  };
  uint32_t edge16 = edge_bits(m0);

  // ---- per-lane LDS byte offsets, constant over the K loop ----
  // ab[tap]: activation fragment of pixel block 0, K sub-step 0 of tap (kr, ks), relative to a window buffer: tile row wr*64 + i32
  // reads window row r + W ky + kx (forward: (ky, kx) = (kr, ks); data gradient: (2 - kr, 2 - ks)); the 16-byte chunk hq + 2 s sits
  // at the position XOR-ed with that window row's key.  (Sub-step s is the same address with bits 5-6 XOR-ed by s: the chunk index
  // is hq | 2 s, disjoint bits; pixel block 1 is 32 rows = a ds_read immediate further on, same key.)
  uint32_t ab[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int kr = t / 3, ks = t % 3;
    const int ky = MODE == 0 ? kr : 2 - kr, kx = MODE == 0 ? ks : 2 - ks;
    const int rr = i32 + W * ky + kx;  // (wr * 64 is a multiple of 16: the key only looks at row bits 0-3)
    const int key = (rr & 7) ^ ((rr >> 3) & 1);
    ab[t] = (uint32_t)((wr * 64 + rr) * 128 + ((hq ^ key) << 4));
  }
  // zs[pb]: the zero PAIR (the last two rows of a window buffer: 256 bytes at a 256-byte boundary = every LDS bank once), minus
  // the pb * 32 rows the read's immediate adds; an edge lane reads the zeros at its own position modulo 256 (conflict-free)
  uint32_t zs[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) zs[f] = (uint32_t)((wrows - 2) * 128 - f * 32 * 128);
  // bb0: weight fragment of channel block 0, sub-step 0, relative to a weight tile: MFMA row i32 <- weight row n(i32) of the wave's
  // 64-channel half (the permutation that leaves 16 consecutive channels in a lane's accumulator registers)
  const int nrow = wc * (BN / WN) + 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
  const uint32_t bb0 = (uint32_t)(nrow * 128 + ((hq ^ ((nrow >> 1) & 7)) << 4));

  const int cpk = p.ci / 64;   // channel chunks (even: ci is a multiple of 128); K order (chunk, tap), tap fastest
  const int n_it = cpk >> 1;   // iterations of the 18-step body
  const int tap_k = cpk * 128;  // byte distance between the weight K offsets of consecutive taps of one chunk

  u32x4 fa[2][4][2], fb[2][4][2];  // [register set][K sub-step s][pixel block / channel block]

  // fragments of the K-step with tap TAP_ in window buffer WBUF_ / weight buffer BBUF_ -> register set SET
#define VDQN_LOAD_FRAGS(SET, TAP_, WBUF_, BBUF_)                                                                         \
  {                                                                                                                      \
    constexpr int kr_ = (TAP_) / 3, ks_ = (TAP_) % 3;                                                                    \
    constexpr int ky_ = MODE == 0 ? kr_ : 2 - kr_, kx_ = MODE == 0 ? ks_ : 2 - ks_;                                      \
    constexpr uint32_t tb_ = (ky_ == 0 ? 1u : 0u) | (ky_ == 2 ? 2u : 0u) | (kx_ == 0 ? 4u : 0u) | (kx_ == 2 ? 8u : 0u);   \
    const unsigned char* wb_ = smem + kM_WinBase + (WBUF_)*kM_WinStride;                                                 \
    _Pragma("unroll") for (int f_ = 0; f_ < 2; ++f_) {                                                                   \
      uint32_t a0_ = ab[TAP_];                                                                                           \
      if constexpr (tb_ != 0u) { /* an edge lane's tap leaves the image: read the zero pair */                           \
        const bool z_ = (edge16 & (tb_ << (4 * f_))) != 0u;                                                              \
        a0_ = z_ ? ((a0_ & 255u) | zs[f_]) : a0_;                                                                        \
      }                                                                                                                  \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                                   \
        fa[SET][s_][f_] = *reinterpret_cast<const u32x4*>(wb_ + f_ * 32 * 128 + (a0_ ^ (uint32_t)(s_ << 5)));            \
    }                                                                                                                    \
    const unsigned char* bt_ = smem + (BBUF_)*kM_WtTile;                                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                     \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                                   \
        fb[SET][s_][j_] = *reinterpret_cast<const u32x4*>(bt_ + j_ * 32 * 128 + (bb0 ^ (uint32_t)(s_ << 5)));            \
  }
#define VDQN_MFMA_ALL(SET)                                                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) _Pragma("unroll") for (int f_ = 0; f_ < 2; ++f_)                     \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                                 \
    acc[f_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[SET][s_][j_]),                   \
                                                          __builtin_bit_cast(bf16x8, fa[SET][s_][f_]), acc[f_][j_], 0, 0, 0); \
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
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                                     \
      asm volatile("" : "+v"(fa[cur_][s_][0]), "+v"(fa[cur_][s_][1]), "+v"(fb[cur_][s_][0]), "+v"(fb[cur_][s_][1]));     \
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
    VDQN_LOAD_FRAGS(nxt_, tl_, cl_ & 1, nxt_) /* unconditional: the step behind the last one re-reads buffers that still exist */ \
    VDQN_MFMA_ALL(cur_)                                                                                                  \
    VDQN_INTERLEAVE32()                                                                                                  \
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
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][j][r] = 0.f;
    VDQN_LOAD_FRAGS(0, 0, 0, 0)  // the fragments of step 0
    // the next tile of this workgroup (if any): where its first window and weight tiles come from
    const uint32_t lt_nx = lt + x_blocks;
    const bool has_nx = lt_nx < x_count;
    const int tn_nx = has_nx ? (int)((x_first + lt_nx) % (uint32_t)p.tiles_n) : tile_n;
    const int tm_nx = has_nx ? (int)((x_first + lt_nx) / (uint32_t)p.tiles_n) : tile_m;
    const int q0_t = tm_nx * BM - W - 1 + lrow;
    const uint32_t b_t = (uint32_t)(tn_nx * BN + lrow) * (uint32_t)(p.ktot * 2) + (uint32_t)(lchunk_b * 16);
    const i32x4 rs_t = tm_nx * BM >= m_split ? rs_b1 : rs_b0;  // the next tile's weight set
#pragma clang loop unroll(disable)
    for (int it = 0; it < n_it; ++it) {
      const int cc2 = 2 * it;            // first chunk of this iteration
      const bool last_it = has_nx && it == n_it - 1;  // (without a next tile the steps behind the end stage this tile's chunk 2 it + 2: nobody reads it)
      const int so_nx = last_it ? 0 : (cc2 + 2) * 128;
      const int q_nx = last_it ? q0_t : q0;
      const uint32_t b_nx = last_it ? b_t : b_off0;
      const i32x4 rs_bn = last_it ? rs_t : rs_b;
      VDQN_USTEP(0) VDQN_USTEP(1) VDQN_USTEP(2) VDQN_USTEP(3) VDQN_USTEP(4) VDQN_USTEP(5) VDQN_USTEP(6) VDQN_USTEP(7) VDQN_USTEP(8)
      VDQN_USTEP(9) VDQN_USTEP(10) VDQN_USTEP(11) VDQN_USTEP(12) VDQN_USTEP(13) VDQN_USTEP(14) VDQN_USTEP(15) VDQN_USTEP(16) VDQN_USTEP(17)
    }
    VDQN_ST(st_comp)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two tiles staged behind the last step have landed (the next tile's
    __builtin_amdgcn_s_barrier();                     // K-steps 0 and 1: weight buffers 0, 1 and window buffer 0 stay untouched)
#ifdef VDQN_STAMP
    st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    // the epilogue's scratch (column sums) is window buffer 1: its last reader was step 16
    win9m_epilogue<BM>(p, acc, smem + kM_WinBase + kM_WinStride, m0, n0, tile_m, rows_total, m0 >= m_split ? p.bias_b : p.bias);
    if (!has_nx) break;
    lt = lt_nx;
    tile_n = tn_nx; tile_m = tm_nx;
    n0 = tile_n * BN; m0 = tile_m * BM;
    q0 = q0_t;
    b_off0 = b_t;
    rs_b = rs_t;
    edge16 = edge_bits(m0);
  }
#undef VDQN_USTEP
#undef VDQN_LOAD_FRAGS
#undef VDQN_MFMA_ALL
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

// entry used by vdqn_conv2d (igemm.hip): returns VDQN_OK or an error code
template <int MODE, int BM>
static void launch_win9m(const IgemmParams& p, hipStream_t stream, void* stamps) {
  using G = Win9mGeom<BM>;
  const int wrows = (BM + 2 * p.wo + 2 + 1 + 7) & ~7;  // <= G::WinRows for W <= 28
  const unsigned tiles = (unsigned)(((p.M + BM - 1) / BM) * p.tiles_n);
  // VDQN_WIN9_PERSIST=1: at most as many workgroups as the chip holds at once, each walking its tiles with the next tile's first
  // two K-steps staged under the current tile's last steps and epilogue; 0: one workgroup per tile
  // (default: for launches of more than two rounds of resident workgroups — layer2 +5-7 %, layer3 at 512 frames +4 %; a launch of
  // 1.5 rounds loses 2-3 % to the static tile assignment; 2: always; profiles/r02o_win9u_persistent.txt)
  static const int persist = [] { const char* e = getenv("VDQN_WIN9_PERSIST"); return e ? atoi(e) : 1; }();
  const unsigned resident = (unsigned)((BM == 128 ? 2 : 1) * vdqn_num_cus());
  const unsigned grid = (BM == 128 && ((persist == 1 && tiles > 2 * resident) || (persist >= 2 && tiles > resident))) ? resident : tiles;  // (256-row tiles: one workgroup per tile)
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9m_kernel<MODE, BM>), (size_t)G::Smem);
  hipLaunchKernelGGL((win9m_kernel<MODE, BM>), dim3(grid), dim3(G::NT), G::Smem, stream, p, wrows, make_fastdiv((uint32_t)p.wo),
                     make_fastdiv((uint32_t)p.howo), tiles, stamps);
}

int vdqn_launch_win9m(const void* pv, int mode, hipStream_t stream) {
  const IgemmParams& p = *reinterpret_cast<const IgemmParams*>(pv);
  // VDQN_WIN9_BM256: 1 = 256-row tiles wherever the launch has more 128-row tiles than the chip holds at once (two per CU),
  // 2 = always, 0 = never
  static const int bm256 = [] { const char* e = getenv("VDQN_WIN9_BM256"); return e ? atoi(e) : 0; }();
  const bool big = bm256 == 2 || (bm256 == 1 && (long long)p.tiles_m * p.tiles_n > 2ll * vdqn_num_cus());
  void* stamps = nullptr;
#ifdef VDQN_STAMP
  stamps = g_stamp_buffer;
#endif
  vdqn_prof_begin(mode == 0 ? "igemm_win<bf16,128,fwd>" : "igemm_win<bf16,128,dgrad>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))), stream);
  if (mode == 0) {
    if (big) launch_win9m<0, 256>(p, stream, stamps); else launch_win9m<0, 128>(p, stream, stamps);
  } else {
    if (big) launch_win9m<1, 256>(p, stream, stamps); else launch_win9m<1, 128>(p, stream, stamps);
  }
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
