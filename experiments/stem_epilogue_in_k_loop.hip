// The ResNet stem for bf16 as one persistent kernel: conv1 7x7/2 (a 4x4/1 convolution over the packed space-to-depth
// operand of vdqn_pack_input) + folded BatchNorm + ReLU + MaxPool2d(3, 2, 1)
// (torchvision resnet.py conv1/bn1/relu/maxpool; archs/HabitatDQNMultiAction.py:30).
//
// Tile = a 16x16 patch of conv pixels (rows 14 ty - 1 .., cols 14 tx - 1 ..) that contains every window of a 7x7 patch
// of pooled pixels; 64 tiles per image.  What makes this kernel different from the generic implicit GEMM (igemm.hip,
// MODE 3, still used for f32):
//   * The operand of a tile is the 19 x 19 packed pixels its conv pixels read through their 4 x 4 taps, staged ONCE as they lie
//     in the image (11.3 KB, three LDS-DMA pieces per thread; round 3 staged a 128-byte K row per conv pixel and kernel row:
//     38 KB, ten pieces, whose issue alone held a wave as long as its K loop).  A fragment read is 16 bytes of window row
//     f + kr at pixel i16 + kx: window row R serves every (conv row f, kernel row kr) pair with f + kr = R — 14 fragment reads
//     per lane and tile, read one row ahead of the MFMAs that use them; a one-bit chunk swizzle keeps them conflict-free.
//   * Workgroups are persistent (two per CU) and walk the tile list with a stride of gridDim.x; the 64 x 256 weight matrix
//     is read ONCE per wave and tile loop into registers (its 32 MFMA fragments = 128 VGPRs) and never touches LDS.
//   * Two window buffers: the LDS-DMA of tile t+1 runs underneath the MFMAs and the pooling epilogue of tile t.
//   * No barrier and no DMA inside a tile's K loop.
// Epilogue, tiles that need arg-max bytes (frames that see a backward pass), as in igemm.hip MODE 3: bias + ReLU -> bf16 patch in
// LDS (a region of its own) -> 49 pooled pixels x 64 channels with the first-maximum-wins rule of maxpool_fwd_kernel.  Tiles
// without arg-max (two thirds of a TD update's frames) are pooled straight from the accumulators: column maxima by DPP row
// shifts, row maxima inside the lane, one pooled row per wave pair exchanged through LDS; the first stages of this epilogue are
// issued among the MFMAs of the K loop's last rows (a conv row is complete three window rows before the loop ends).  A workgroup walks its
// arg-max tiles and then its plain tiles in two loops over one body, so that each instance is register-allocated with its own
// epilogue only.  Round 4 (tools/stem_phases.py, tools/stamp_stem.py): stem_conv_pool 0.446 -> 0.398 ms per update
// with the register pooling (profiles/r04at_*), 0.473 -> 0.420 on a slower box with the compact window on top (r04bd);
// -DVDQN_STEM_LDS_POOL sends every tile through the LDS epilogue.
// Results are bit-identical to vdqn_conv2d followed by vdqn_maxpool_fwd (same K order, same rounding points).
#include "common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

struct StemParams {
  const bf16raw* t_in;   // [n][115][115][16]
  const bf16raw* wt;     // [64][256]
  const float* bias;     // [64]
  bf16raw* pool;         // [n][56][56][64]
  uint8_t* idx;          // [n][56][56][64]
  int n_img, n_tiles;
  int n_idx_img;         // images [0, n_idx_img) get arg-max bytes
  void* stamps;          // -DVDQN_STAMP builds only (tools/stamp_stem.py): s_memtime at the phase boundaries of a workgroup's first tiles
};

struct KindArg { static constexpr bool value = true; };     // tile loop instances of stem_kernel
struct KindPlain { static constexpr bool value = false; };

// LDS: two windows of 19 x 19 packed pixels (32 B each, stored as they lie in the image: 38 chunks of 16 B per row, 722 chunks
// rounded up to three staging pieces of 256), the arg-max path's bf16 patch, the bias vector and the row-exchange slots of the
// plain path.  59 KiB: two workgroups per CU (the weights live in registers).
constexpr int kRowPitch = 38 * 16;             // bytes per window row
constexpr int kWBytes = 3 * 256 * 16;          // one window buffer
constexpr int kPatchOff = 2 * kWBytes;
constexpr int kBiasOff = kPatchOff + 256 * 128;
constexpr int kXchOff = kBiasOff + 256;        // 4 sender waves x 64 lanes x 32 B (every lane writes: no branch inside the K loop)
constexpr int kSmem = kXchOff + 4 * 64 * 32;
constexpr unsigned kOobS = 0x80000000u;

// Images from p.n_idx_img on get no arg-max bytes (frames that never see a backward pass: the s' half of the online pass and the
// whole target pass of a TD update): their pooling is a plain packed 16-bit maximum, a quarter of the vector instructions.
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;                    // [2][19 rows][38 chunks][16 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_wave = lds_base + (uint32_t)wave_u * 1024u;
  constexpr int PSTR = 256 * 16;  // bytes per staging piece
  constexpr long long kImgBytes = 115ll * 115 * 16 * 2;

  // ---- weights: this wave's fragments of all four K-steps in REGISTERS (128 VGPRs): fragment j of K-step kr, K half h = 16 bytes
  // of weight row (i16 >> 2) * 16 + j * 4 + (i16 & 3) (the permuted order of igemm.hip).  Loaded once per tile LOOP (there are
  // two, below): as one value live through both loops the register allocator spilled two fragments to scratch and reloaded them
  // inside every K loop; loaded again in front of the second loop they are two independent live ranges and nothing spills. ----
  auto load_weights = [&](u32x4 (&fb)[4][2][4]) {
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.wt) + (size_t)((lane & 15) >> 2) * 16 * 512 + (size_t)(lane & 3) * 512;
    asm volatile("" : "+v"(wrow));  // (a pointer the optimiser cannot match with the other loop's: two sets of loads)
#pragma unroll
    for (int kr = 0; kr < 4; ++kr)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fb[kr][h][j] = *reinterpret_cast<const u32x4*>(wrow + (size_t)j * 4 * 512 + kr * 128 + ((lane >> 4) + 4 * h) * 16);
  };
  // Stage the window of tile t into buffer `buf`: the 19 x 19 packed pixels (y0 .. y0 + 18, x0 .. x0 + 18) a 16 x 16 patch of
  // conv pixels reads with its 4 x 4 taps, ONCE each (11.3 KB, three LDS-DMA pieces per thread).  Round 3 staged one 128-byte
  // K row (four neighbouring pixels) per conv pixel and kernel row — 38 KB and ten pieces per tile for the same 11.3 KB of image,
  // and the ten pieces held each wave for ≈ 2200 cycles per tile, as long as the K loop (tools/stamp_stem.py,
  // profiles/r04aj_stamp_stem_before.txt).  The LDS image of a piece is lane-linear, so the swizzle (chunk c of a row stored at
  // c ^ ((c >> 4) & 1): the two halves of a pixel swap places in pixels 8..15) is applied to the SOURCE chunk.
  auto issue_window = [&](int t, int buf) {  // t: logical tile id (tile_at)
    int tid_w = tid;
    asm volatile("" : "+v"(tid_w));  // (a thread's staging slots are recomputed per tile instead of living in registers across the K loop)
    const int img = t >> 6, ty = (t >> 3) & 7, tx = t & 7;
    const int y0 = 14 * ty - 1, x0 = 14 * tx - 1;
    const unsigned long long a_ptr = (unsigned long long)((const unsigned char*)p.t_in + (long long)img * kImgBytes);
    const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)), (int)kImgBytes, 0x00020000};
    uint32_t vw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int sl = i * 256 + tid_w;            // slot = window row * 38 + stored chunk
      const int row = (sl * 1725) >> 16;        // sl / 38, exact for sl < 768
      const int pc = sl - row * 38;
      const int lc = pc ^ ((pc >> 4) & 1);       // the image chunk this slot holds
      const int sy = y0 + row;
      // packed column -1 / 115 (only read for conv columns that are never pooled) wraps inside the image or falls out of the
      // descriptor's range (zeros): either way harmless; rows outside the image and the slots behind row 18 are zero-filled
      const bool ok = row < 19 && (unsigned)sy < 115u;
      vw[i] = ok ? (uint32_t)((sy * 115 + x0) * 32 + lc * 16) : kOobS;
    }
    const uint32_t l0 = lds_wave + (uint32_t)(buf * kWBytes);
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %4, 0 offen lds\n\t"
        "s_add_u32 m0, %3, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds\n\t"
        "s_add_u32 m0, %3, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds"
        ::"v"(vw[0]), "v"(vw[1]), "v"(vw[2]), "s"(l0), "s"(rs_a), "n"(PSTR), "n"(2 * PSTR)
        : "memory", "scc");
  };
  if (tid < 64) reinterpret_cast<float*>(smem + kBiasOff)[tid] = p.bias[tid];  // visible after the first tile's barrier

  // fragment read of conv pixel (4 wave + f, i16), kernel row kr, K half h: 16 bytes = chunk 2 i16 + 4 h + g of window row
  // 4 wave + f + kr (pixel i16 + 2 h + g / 2, channel half g & 1), at its swizzled place.  The 16 lanes of a group read chunks
  // c, c + 2, ..., c + 30: lanes 8 apart would meet in one bank quad, the swizzle puts them in neighbouring ones.
  const int lc0 = 2 * i16 + g, lc1 = lc0 + 4;
  const int coff0 = (lc0 ^ ((lc0 >> 4) & 1)) << 4, coff1 = (lc1 ^ ((lc1 >> 4) & 1)) << 4;

  // The workgroup's tile sequence: first its tiles WITH arg-max bytes (images < n_idx_img), then those without, as two loops over
  // the same body — each instance is compiled with its own epilogue only, so the loop invariants of one pooling path do not
  // occupy registers in the other's K loop (one common loop spilled 13 VGPRs and reloaded them in front of every window issue:
  // three scratch round trips per tile).  Every workgroup gets the same share of both kinds (tiles b, b + G, b + 2G, ... of
  // either range), and inside a range the ids are remapped so that the workgroups of one XCD (blockIdx % 8; G is a multiple of 8
  // or the whole tile count) walk a contiguous piece: neighbouring patches share two packed rows / columns, which then hit in
  // that XCD's L2.
#ifdef VDQN_STEM_LDS_POOL
  const int Ta = p.n_tiles;        // every tile through the LDS-patch epilogue
#else
  const int Ta = p.n_idx_img * 64;
#endif
  const int Tn = p.n_tiles - Ta, G = (int)gridDim.x, b0 = (int)blockIdx.x;
  const int ka = b0 < Ta ? (Ta - b0 + G - 1) / G : 0;
  const int kn = b0 < Tn ? (Tn - b0 + G - 1) / G : 0;
  auto tile_at = [&](int k) {  // logical id of this workgroup's k-th tile; -1 behind the last
    if (k < ka) return (int)xcd_remap((uint32_t)(b0 + k * G), (uint32_t)Ta);
    if (k < ka + kn) return Ta + (int)xcd_remap((uint32_t)(b0 + (k - ka) * G), (uint32_t)Tn);
    return -1;
  };
  auto tile = [&](auto kind, const int k, const u32x4 (&fb)[4][2][4]) {
    constexpr bool ARG = decltype(kind)::value;
    const int buf = k & 1;
    const int tl = tile_at(k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // window t visible; everyone is done pooling the previous tile (its patch lived in buf ^ 1)
#ifdef VDQN_STAMP
    // row of 8 words per (workgroup, tile number < 16): top, DMA issued, K loop done, patch written, pooled, HW_ID
    const int st_k = k;
    unsigned long long* st_row = (p.stamps && st_k < 16) ? reinterpret_cast<unsigned long long*>(p.stamps) + ((size_t)blockIdx.x * 16 + st_k) * 8 : nullptr;
    if (st_row && tid == 0) { st_row[0] = __builtin_amdgcn_s_memtime(); st_row[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); }
#endif
    // (The ten pieces of the burst hold the wave for ≈ 2200 cycles, as long as the K loop itself, while the CU's other workgroup
    // has the matrix pipe: a CU's LDS-DMA queue drains one piece per wave and ≈ 220 cycles.  Issuing the pieces between the MFMA
    // groups of the K loop, or spread over the whole tile, moves that wait into the K loop and leaves the tile period where it
    // was — tools/stamp_stem.py, profiles/r04aj-r04al_*, experiments/README.md.)
    {
      const int nx = tile_at(k + 1);
      if (nx >= 0) issue_window(nx, buf ^ 1);
    }
#ifdef VDQN_STAMP
    if (st_row && tid == 0) st_row[1] = __builtin_amdgcn_s_memtime();
#endif

    // ---- 4 K-steps straight out of LDS: acc[f][j] = pixels (4 wave + f, i16), channels g*16 + j*4 + reg ----
    f32x4 acc[4][4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int img = tl >> 6, ty = (tl >> 3) & 7, tx = tl & 7;
    // ---- helpers of the plain (no arg-max) epilogue; its first stages run INSIDE the K loop (below) ----
    int lane_l = lane;
    asm volatile("" : "+v"(lane_l));  // (lane-derived addresses of this path are recomputed per tile instead of living in registers across the K loop)
    const int i16p = lane_l & 15, gp = lane_l >> 4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const uint32_t keep = (tx == 0 && i16p == 0) ? 0u : 0xffffffffu;  // conv column -1: this lane's own values drop out
    // (the bias is read from LDS eight channels at a time, where it is used: sixteen resident values are registers the K loop of
    // this instance does not have)
    auto bias8 = [&](int hf, float (&b8)[8]) {
      *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(smem + kBiasOff + (gp * 16 + hf * 8) * 4);
      *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(smem + kBiasOff + (gp * 16 + hf * 8 + 4) * 4);
    };
    auto cvt = [&](int f, int e, const float (&b8)[8]) {  // channels 2e, 2e + 1 of patch row f: bias, bf16, packed (v_pk_add_f32, v_cvt_pk_bf16_f32)
      const f32x2 v = f32x2{acc[f][e >> 1][(e & 1) * 2], acc[f][e >> 1][(e & 1) * 2 + 1]} + f32x2{b8[2 * (e & 3)], b8[2 * (e & 3) + 1]};
      return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    };
    auto pmax = [](uint32_t a, uint32_t b) {
      return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b)));
    };
    auto hmax = [&](uint32_t x) {  // max over lanes i16, i16 + 1, i16 + 2 (lanes past the row read 0 = +0.0; only odd / unused lanes see them)
      const uint32_t x1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true);  // row_shl:1
      const uint32_t x2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x102, 0xf, 0xf, true);  // row_shl:2
      return pmax(pmax(x & keep, x1), x2);
    };
    const bool col_lane = (i16p & 1) == 0 && i16p <= 12;
    // exchange slots: [sender wave][lane][32 B] (wave 0's slot is never read)
    auto xch = [&](int sender) { return smem + kXchOff + (sender * 64 + lane_l) * 32; };
    const bool row_out = ty == 0 && wave_u == 0;  // conv row -1: patch row 0 of the first tile row drops out (uniform)
    uint32_t P0[8], Q[8], P2[8];  // plain instance: packed rows / running maxima carried from the K loop into the epilogue
    // Conv row f is complete after window row f + 3, so its bias / rounding / column maxima are issued among the MFMAs of the rows
    // that follow (three vector instructions in the shadow of every MFMA) instead of behind the K loop:
    //   stage 0 (among row 4's MFMAs): P0 = row 0 packed; its column maximum goes to the exchange slot of the wave above
    //   stage 1 (row 5): Q = max(P0, row 1);   stage 2 (row 6): P2 = row 2, Q = max(Q, P2)
    auto early = [&](int stage) {
      if (stage == 0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float b8[8];
          bias8(hf, b8);
#pragma unroll
          for (int e = 0; e < 4; ++e) P0[hf * 4 + e] = row_out ? 0u : cvt(0, hf * 4 + e, b8);
        }
        uint32_t S[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) S[e] = hmax(P0[e]);
        unsigned char* d = xch(wave_u);  // (every lane of every wave: straight-line code that the MFMAs of row 4 can be issued among)
        *reinterpret_cast<uint4*>(d) = make_uint4(S[0], S[1], S[2], S[3]);
        *reinterpret_cast<uint4*>(d + 16) = make_uint4(S[4], S[5], S[6], S[7]);
      } else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float b8[8];
          bias8(hf, b8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (stage == 1) Q[hf * 4 + e] = pmax(P0[hf * 4 + e], cvt(1, hf * 4 + e, b8));
            else {
              P2[hf * 4 + e] = cvt(2, hf * 4 + e, b8);
              Q[hf * 4 + e] = pmax(Q[hf * 4 + e], P2[hf * 4 + e]);
            }
          }
        }
      }
    };
    const unsigned char* a_rd = sW + buf * kWBytes + (4 * wave) * kRowPitch;
#if !(defined(VDQN_STEM_PROBE) && (VDQN_STEM_PROBE & 1))  // diagnostic builds (tools/stem_phases.py): bit 0 = no K loop
    // Window row R = f + kr serves every (conv row f, kernel row kr) pair on its diagonal: 14 fragment reads per tile instead of
    // 32.  For a fixed f the products still arrive in the order kr = 0..3, h = 0, 1 — the accumulation order of vdqn_conv2d.
    // (the two fragments of row R + 1 are read before the MFMAs of row R are issued: one LDS latency per tile instead of seven)
    u32x4 fa[2][2];
    fa[0][0] = *reinterpret_cast<const u32x4*>(a_rd + coff0);
    fa[0][1] = *reinterpret_cast<const u32x4*>(a_rd + coff1);
#pragma unroll
    for (int R = 0; R < 7; ++R) {
      if (R < 6) {
        fa[(R + 1) & 1][0] = *reinterpret_cast<const u32x4*>(a_rd + (R + 1) * kRowPitch + coff0);
        fa[(R + 1) & 1][1] = *reinterpret_cast<const u32x4*>(a_rd + (R + 1) * kRowPitch + coff1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!ARG) {
        if (R >= 4) early(R - 4);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int kr = (R > 3 ? R - 3 : 0); kr <= (R < 3 ? R : 3); ++kr) {
          const int f = R - kr;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[kr][h][j]), __builtin_bit_cast(bf16x8, fa[R & 1][h]), acc[f][j], 0, 0, 0);
        }
      }
      if constexpr (!ARG) {
        if (R >= 4) {
#pragma unroll
          for (int i = 0; i < 8 * (7 - R); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // three vector instructions of the early stage
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // (keeps the next row's reads in front of this row's MFMAs and the rows in order)
    }
#else
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[f][j][0] = __builtin_bit_cast(float, fb[f][0][j][0] & 0x3f800000u);  // (keeps the weight registers live)
    (void)a_rd; (void)coff0; (void)coff1;
#endif

    // ---- bias + ReLU -> bf16 patch in LDS (over the window just consumed), then the 7x7 pooled pixels ----
    // (The pooling reads walk the patch with a stride of TWO rows and meet only half of the LDS banks: 36 % of this kernel's LDS
    // cycles are bank conflicts.  Round 3 tried the layout the bank model prefers — rows pair-swapped, another chunk key: 684 -> 288
    // model cycles per tile — and the kernel went from 0.49 to 0.87 ms per update: its limit is VALU issue (≈ 650 vector
    // instructions per tile and wave beside 128 MFMAs; SQ_ACTIVE_INST_VALU is the largest share), and the per-access key
    // arithmetic costs more issue slots than the conflicts cost LDS cycles.  Kept as it was.)
#if defined(VDQN_STEM_PROBE) && (VDQN_STEM_PROBE & 4)  // bit 2 = no bias / ReLU / patch phase either (the accumulators stay live)
    {
      float sum = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[f][j][0] + acc[f][j][1] + acc[f][j][2] + acc[f][j][3];
      if (sum == 12345.678f) p.pool[tid] = 1;
      return;
    }
#endif
#ifdef VDQN_STAMP
    if (st_row && tid == 0) { asm volatile("s_nop 0" ::"v"(acc[3][3][3])); st_row[2] = __builtin_amdgcn_s_memtime(); }
#endif
    if constexpr (!ARG) {
      // ---- no arg-max: pooled straight from the accumulators.  A lane holds conv-patch rows 4 wave + f (f = 0..3), column i16,
      // 16 channels: the three columns of a pooling window are the lanes i16, i16 + 1, i16 + 2 of a 16-lane row (two DPP row
      // shifts; the even lanes 0..12 end with pooled column i16 / 2), the three rows of pooled row 2 wave are the lane's own
      // f = 0..2, and pooled row 2 wave + 1 needs rows f = 2, 3 and row f = 0 of the NEXT wave — the only thing that goes through
      // LDS (its column maximum), behind the tile's
      // only barrier besides the one at its top.  No patch in LDS, no patch barriers, no strided patch reads.
      // bf16 bit patterns of non-negative values order like SIGNED 16-bit integers and every negative value (-0.0 included) is a
      // negative integer, so the ReLU is folded into the maxima: max(+0.0, a, b, ...) over the unclamped values.  Taps outside
      // the image (conv row / column -1: patch row 0 of the first tile row, this lane's own column in lane 0 of the first tile
      // column) are replaced by +0.0, which is what a maximum that starts at +0.0 and skips them gives.  Same bits as the LDS
      // path below and as the two separate kernels (rounding and ReLU are monotonic: they commute with the maximum).
      // (Register budget: the weights hold 128 VGPRs, so patch row f = 0 is converted first — the only row the exchange needs —
      // and rows 1..3 are converted, pooled and stored in two channel halves behind the barrier.)
      uint32_t Bv[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float b8[8];
        bias8(hf, b8);
#pragma unroll
        for (int e = 0; e < 4; ++e) Bv[hf * 4 + e] = pmax(P2[hf * 4 + e], cvt(3, hf * 4 + e, b8));
      }
      __syncthreads();
#ifdef VDQN_STAMP
      if (st_row && tid == 0) st_row[3] = __builtin_amdgcn_s_memtime();
#endif
      const size_t o0 = (((size_t)img * 56 + 7 * ty + 2 * wave_u) * 56 + 7 * tx + (i16p >> 1)) * 64 + gp * 16;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        uint32_t A[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) A[e] = pmax(hmax(Q[hf * 4 + e]), 0u);
        if (col_lane) *reinterpret_cast<uint4*>(p.pool + o0 + hf * 8) = make_uint4(A[0], A[1], A[2], A[3]);
        if (wave_u < 3) {
          uint4 n = make_uint4(0, 0, 0, 0);
          if (col_lane) n = *reinterpret_cast<const uint4*>(xch(wave_u + 1) + hf * 16);
          const uint32_t N[4] = {n.x, n.y, n.z, n.w};
          uint32_t B[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) B[e] = pmax(pmax(hmax(Bv[hf * 4 + e]), N[e]), 0u);
          if (col_lane) *reinterpret_cast<uint4*>(p.pool + o0 + 56 * 64 + hf * 8) = make_uint4(B[0], B[1], B[2], B[3]);
        }
      }
#ifdef VDQN_STAMP
      if (st_row && tid == 0) st_row[4] = __builtin_amdgcn_s_memtime();
#endif
      return;
    }
    float bv[16];  // (re-read per tile from LDS: the weights occupy the registers a resident copy would need)
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<float4*>(bv + 4 * e) = *reinterpret_cast<const float4*>(smem + kBiasOff + (g * 16 + 4 * e) * 4);
    // (the patch has a region of its own: everyone finished pooling the previous tile's patch before this tile's top barrier)
    bf16raw* sT = reinterpret_cast<bf16raw*>(smem + kPatchOff);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int r = wave * 64 + f * 16 + i16;
      bf16raw ov[16];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) ov[j * 4 + q] = f32_to_bf16(fmaxf(acc[f][j][q] + bv[j * 4 + q], 0.f));
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int chunk = (g * 2 + c) ^ (r & 7);
        *reinterpret_cast<uint4*>(sT + (size_t)r * 64 + chunk * 8) = reinterpret_cast<const uint4*>(ov)[c];
      }
    }
    __syncthreads();
#ifdef VDQN_STAMP
    if (st_row && tid == 0) st_row[3] = __builtin_amdgcn_s_memtime();
#endif
#if defined(VDQN_STEM_PROBE) && (VDQN_STEM_PROBE & 2)  // bit 1 = no pooling phase (one word per lane keeps the patch writes live)
    if (tid == 0) p.pool[(size_t)tl * 64] = sT[0];
    if (false)
#endif
    for (int item = tid; item < 49 * 8; item += 256) {
      const int pp = item >> 3, cg = item & 7;
      const int pi = pp / 7, pj = pp - pi * 7;
      const size_t o = (((size_t)img * 56 + 7 * ty + pi) * 56 + 7 * tx + pj) * 64 + cg * 8;
      if (img >= p.n_idx_img) {  // (uniform over the workgroup)
        // no arg-max: post-ReLU bf16 bit patterns order like SIGNED 16-bit integers, and the -0.0 the ReLU may leave (0x8000, the
        // smallest of them) loses against the initial +0.0 exactly as the arg-max path's sign mask drops it
        const bool r0 = ty == 0 && pi == 0, c0 = tx == 0 && pj == 0;
        i16x2 m[4] = {i16x2{0, 0}, i16x2{0, 0}, i16x2{0, 0}, i16x2{0, 0}};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const bool out = (kh == 0 && r0) || (kw == 0 && c0);  // such a tap re-reads the centre tap instead (always inside)
            const int r = out ? (2 * pi + 1) * 16 + 2 * pj + 1 : (2 * pi + kh) * 16 + 2 * pj + kw;
            const uint4 v = *reinterpret_cast<const uint4*>(sT + (size_t)r * 64 + ((cg ^ (r & 7)) * 8));
            m[0] = __builtin_elementwise_max(m[0], __builtin_bit_cast(i16x2, v.x));
            m[1] = __builtin_elementwise_max(m[1], __builtin_bit_cast(i16x2, v.y));
            m[2] = __builtin_elementwise_max(m[2], __builtin_bit_cast(i16x2, v.z));
            m[3] = __builtin_elementwise_max(m[3], __builtin_bit_cast(i16x2, v.w));
          }
        uint4 ov;
        ov.x = __builtin_bit_cast(uint32_t, m[0]); ov.y = __builtin_bit_cast(uint32_t, m[1]);
        ov.z = __builtin_bit_cast(uint32_t, m[2]); ov.w = __builtin_bit_cast(uint32_t, m[3]);
        *reinterpret_cast<uint4*>(p.pool + o) = ov;
        continue;
      }
      // Straight-line: all nine taps are read up front (one exposed LDS latency per item instead of nine).  Only the taps of
      // conv row / column -1 can be outside the image (first tile row / column, first pooled row / column): they are read too
      // (the patch row exists) and masked out of the maximum.  The centre tap is always valid, so every key ends up tagged.
      const uint32_t row0 = (ty == 0 && pi == 0) ? 0u : 0x7fff0u, col0 = (tx == 0 && pj == 0) ? 0u : 0x7fff0u;
      uint4 v[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int r = (2 * pi + kh) * 16 + 2 * pj + kw;
          v[kh * 3 + kw] = *reinterpret_cast<const uint4*>(sT + (size_t)r * 64 + ((cg ^ (r & 7)) * 8));
        }
      // post-ReLU bf16 bit patterns order like unsigned integers: key = bits << 4 | (8 - tap), one v_max_u32 per element and tap
      uint32_t key[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) key[e] = 0u;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const uint32_t m = (kh == 0 ? row0 : 0x7fff0u) & (kw == 0 ? col0 : 0x7fff0u);
          const uint32_t tag = (kh == 0 || kw == 0) ? ((m >> 4) & (uint32_t)(8 - (kh * 3 + kw))) : (uint32_t)(8 - (kh * 3 + kw));
          const uint32_t w4[4] = {v[kh * 3 + kw].x, v[kh * 3 + kw].y, v[kh * 3 + kw].z, v[kh * 3 + kw].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            key[2 * q] = max(key[2 * q], ((w4[q] << 4) & m) | tag);
            key[2 * q + 1] = max(key[2 * q + 1], ((w4[q] >> 12) & m) | tag);
          }
        }
      // key >> 4 IS the bf16 bit pattern of the maximum; 8 - (key & 15) the tap that holds it (first maximum wins)
      uint4 ov;
      uint2 bi;
      ov.x = (key[0] >> 4) | ((key[1] << 12) & 0xffff0000u);
      ov.y = (key[2] >> 4) | ((key[3] << 12) & 0xffff0000u);
      ov.z = (key[4] >> 4) | ((key[5] << 12) & 0xffff0000u);
      ov.w = (key[6] >> 4) | ((key[7] << 12) & 0xffff0000u);
      bi.x = (8u - (key[0] & 15u)) | ((8u - (key[1] & 15u)) << 8) | ((8u - (key[2] & 15u)) << 16) | ((8u - (key[3] & 15u)) << 24);
      bi.y = (8u - (key[4] & 15u)) | ((8u - (key[5] & 15u)) << 8) | ((8u - (key[6] & 15u)) << 16) | ((8u - (key[7] & 15u)) << 24);
      *reinterpret_cast<uint4*>(p.pool + o) = ov;
      *reinterpret_cast<uint2*>(p.idx + o) = bi;
    }
#ifdef VDQN_STAMP
    if (st_row && tid == 0) st_row[4] = __builtin_amdgcn_s_memtime();
#endif
  };
  if (ka + kn > 0) issue_window(tile_at(0), 0);
  if (ka > 0) {
    u32x4 fb[4][2][4];
    load_weights(fb);
    for (int k = 0; k < ka; ++k) tile(KindArg{}, k, fb);
  }
  if (kn > 0) {
    u32x4 fb[4][2][4];
    load_weights(fb);
    for (int k = ka; k < ka + kn; ++k) tile(KindPlain{}, k, fb);
  }
}

}  // namespace

// bf16 entry used by vdqn_stem_conv_pool / vdqn_stem_conv_pool_n (igemm.hip); arg-max bytes for the first n_idx_img images only
// (idx == nullptr: none); returns VDQN_OK or an error code
int vdqn_stem_bf16(const void* t_in, const void* wt, const float* bias, void* pool, void* idx, int n_img, int n_idx_img, hipStream_t st) {
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&stem_kernel), (size_t)kSmem);
  const int n_cu = vdqn_num_cus();
  StemParams p;
  p.t_in = (const bf16raw*)t_in; p.wt = (const bf16raw*)wt; p.bias = bias; p.pool = (bf16raw*)pool; p.idx = (uint8_t*)idx;
  p.n_img = n_img;
  p.n_idx_img = idx ? n_idx_img : 0;
  p.n_tiles = n_img * 64;
  p.stamps = nullptr;
#ifdef VDQN_STAMP
  { extern void* g_stamp_buffer; p.stamps = g_stamp_buffer; }
#endif
  const int grid = p.n_tiles < 2 * n_cu ? p.n_tiles : 2 * n_cu;
  vdqn_prof_begin("stem_conv_pool<bf16>", 2.0 * n_img * 112 * 112 * 64 * 147,
                  2.0 * ((double)n_img * 115 * 115 * 16 + 64.0 * 256 + (double)n_img * 56 * 56 * 64) + (double)p.n_idx_img * 56 * 56 * 64, st);
  hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(256), kSmem, st, p);
  vdqn_prof_end(st);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
