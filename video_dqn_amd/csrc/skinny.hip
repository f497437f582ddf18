// Skinny GEMMs of the Q-head (bf16): features.8 (3x3 valid convolution over the 7x7x512 trunk output) and the `top` MLP
// Linear(1600F,512)-ReLU-Linear(512,256)-ReLU-Linear(256,15), forward and data gradient
// (archs/HabitatDQNMultiAction.py:30-31,52-54; their autograd backward behind loss.backward(), train_q_network.py:226).
//
// These layers have 256-512 output rows (samples) or 12,800 (features.8) and K = 64 ... 4608: on the generic implicit GEMM
// (128-row tiles, one K-step per barrier) they are 4-100 workgroups walking 25-72 dependent K-steps each — ten launches of
// 8-45 us with the chip nearly empty (~280 us of every update).  What bounds such a GEMM is the latency of the
// K walk and how many CUs pull operand bytes at once, not the matrix pipe.  So here:
//   * small tiles (32 x 32 for the linear layers, 64 x 64 for features.8): 32-400 workgroups per launch instead of 4-100;
//   * the K range of a tile is SPLIT OVER THE FOUR WAVES of its workgroup (wave w takes the 32-deep chunks w, w + 4, ...):
//     a wave's serial chain is a quarter as long, the four partial tiles are added through LDS — inside the workgroup, no
//     cross-workgroup hand-over (the split-K of DESIGN.md 3e lost to its release / acquire pairs);
//   * no LDS staging and no barrier in the K walk: every lane loads its MFMA fragment straight from global memory (operand rows
//     are K-contiguous, a 16-byte load per lane is exactly the lane's eight k values of v_mfma_f32_16x16x32_bf16), kept DEPTH
//     chunks ahead in registers;
//   * one epilogue over the reduced tile: bias, ReLU, the ReLU mask of a data gradient, bf16 store, optional f32 copy (Q) and
//     per-tile column sums (the bias gradient of the layer below) at this kernel's own row granularity (kSkinnyPartRows).
// Arithmetic: products of bf16 operands accumulated in f32 as everywhere else; the K order differs from the generic kernel's
// (four interleaved partial sums), so results agree with it to f32 rounding, not bit for bit.
#include <stdlib.h>

#include "igemm_common.h"

namespace {

// One chunk = 32 consecutive k of every operand row.  Linear layers: row m of A starts at in + m * pix_stride.  CONV (features.8
// forward: r x s valid convolution, stride 1): row m = (img, oy, ox) and chunk c covers tap c * 32 / ci, channels (c * 32) % ci.
// kDepth = chunks in flight per wave; NW = waves per workgroup (the K range is split NW ways)
template <int TM, int TN, bool CONV, int kDepth, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void skinny_kernel(const IgemmParams p, const int n_chunks, const FastDiv d_wo, const FastDiv d_howo) {
  using T = bf16raw;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int PITCH = TN + 4;  // f32 row pitch of a partial tile in LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char skinny_smem[];
  float (*sRed)[TM][PITCH] = reinterpret_cast<float (*)[TM][PITCH]>(skinny_smem);  // [NW waves][TM][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int tile_n = (int)(blockIdx.x % (unsigned)p.tiles_n), tile_m = (int)(blockIdx.x / (unsigned)p.tiles_n);
  const int m0 = tile_m * TM, n0 = tile_n * TN;

  // per-lane operand row bases (bytes); rows beyond M are clamped to the last row (their results are never stored)
  const unsigned char* __restrict__ a_base[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int m = m0 + i * 16 + i16;
    m = m < p.M ? m : p.M - 1;
    long pix = m;
    if constexpr (CONV) {
      const uint32_t img = fastdiv((uint32_t)m, d_howo);
      const uint32_t rem = (uint32_t)m - img * (uint32_t)p.howo;
      const uint32_t oy = fastdiv(rem, d_wo);
      const uint32_t ox = rem - oy * (uint32_t)p.wo;
      pix = ((long)img * p.hi + oy) * p.wi + ox;
    }
    a_base[i] = reinterpret_cast<const unsigned char*>(p.in) + pix * p.pix_stride * 2 + g * 16;
  }
  const unsigned char* __restrict__ w_base[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) w_base[j] = reinterpret_cast<const unsigned char*>(p.wt) + (long)(n0 + j * 16 + i16) * p.ktot * 2 + g * 16;

  // chunk c of this wave = global chunk wave + NW c
  const int my_chunks = (n_chunks - wave + NW - 1) / NW;
  auto a_off = [&](int c) -> long {  // byte offset of global chunk (wave + NW c) inside an A row
    const int k = (wave + NW * c) * 32;
    if constexpr (CONV) {
      const int tap = k / p.ci, c0 = k - tap * p.ci;
      const int ky = tap / p.s, kx = tap - ky * p.s;
      return ((long)(ky * p.wi + kx) * p.pix_stride + c0) * 2;
    } else {
      return (long)k * 2;
    }
  };

  uint4 fa[kDepth][FM], fb[kDepth][FN];
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load = [&](int d, int c) {
    c = c < my_chunks ? c : (my_chunks > 0 ? my_chunks - 1 : 0);  // beyond the end: re-load the last chunk (unused)
    const long ao = a_off(c);
    const long wo = (long)(wave + NW * c) * 64;
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[d][i] = *reinterpret_cast<const uint4*>(a_base[i] + ao);
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[d][j] = *reinterpret_cast<const uint4*>(w_base[j] + wo);
  };

  if (my_chunks > 0) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) load(d, d);
    for (int c = 0; c < my_chunks; c += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        if (c + d < my_chunks) {
          // weights are the first operand: lane (i16, g) ends with output channels 4 g .. 4 g + 3 of fragment j for row i16 of fragment i
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[d][j]), __builtin_bit_cast(bf16x8, fa[d][i]), acc[i][j], 0, 0, 0);
          load(d, c + d + kDepth);
        }
      }
    }
  }
  // partial tiles -> LDS: sRed[wave][row][col], a lane writes four consecutive columns
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(&sRed[wave][i * 16 + i16][j * 16 + g * 4]) = acc[i][j];
  __syncthreads();

  // epilogue: thread -> (row, four consecutive columns)
  constexpr int TPR = TN / 4;          // threads per row
  constexpr int RPP = 64 * NW / TPR;   // rows per pass
  const int er = tid / TPR, ec = (tid % TPR) * 4;
  const int ncol = n0 + ec;
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
  float bv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bv[e] = (p.bias && ncol + e < p.co) ? p.bias[ncol + e] : 0.f;
#pragma unroll
  for (int it = 0; it < TM / RPP; ++it) {
    const int r = it * RPP + er;
    const int m = m0 + r;
    f32x4 v = *reinterpret_cast<const f32x4*>(&sRed[0][r][ec]);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&sRed[w][r][ec]);
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    float x[4];
    const bool ok = m < p.M && ncol < p.co;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[e] = v[e] + bv[e];
      if (p.relu) x[e] = fmaxf(x[e], 0.f);
    }
    if (mask && ok) {
      const uint2 mv = *reinterpret_cast<const uint2*>(mask + (size_t)m * p.ldo + ncol);
      const T* pm = reinterpret_cast<const T*>(&mv);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (to_f32<T>(pm[e]) > 0.f) ? x[e] : 0.f;
    }
    T ov[4];
    float back[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool col_ok = ncol + e < p.co;
      if (!col_ok) x[e] = 0.f;
      ov[e] = from_f32<T>(x[e]);
      back[e] = ok ? to_f32<T>(ov[e]) : 0.f;
    }
    if (ok) {
      if (out) *reinterpret_cast<uint2*>(out + (size_t)m * p.ldo + ncol) = *reinterpret_cast<const uint2*>(ov);
      if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldo + ncol) = make_float4(x[0], x[1], x[2], x[3]);
    }
    if (p.colsum_part) {  // the stored (rounded) values go back for the column sums; own element only: no hazard with other threads
      *reinterpret_cast<f32x4*>(&sRed[0][r][ec]) = f32x4{back[0], back[1], back[2], back[3]};
    }
  }
  if (p.colsum_part) {  // uniform branch
    __syncthreads();
    if (tid < TN && n0 + tid < p.co) {
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < TM; ++r) t += sRed[0][r][tid];
      p.colsum_part[(size_t)tile_m * p.ldo + n0 + tid] = t;
    }
  }
}

// features.8 with its input images in LDS: a workgroup owns G whole images (G * ho * wo <= 64 output pixels, G = 2 for the 7 x 7 -> 5 x 5
// geometry), stages their hi * wi pixels ONCE by LDS-DMA (one 2 ci-byte pixel per DMA piece at a pitch of 2 ci + 32 bytes: the lanes
// of a fragment read then fall on different bank quads) and serves all r * s taps from there — the skinny kernel above re-reads every
// input pixel nine times from beyond L2 (a workgroup's 125 KB working set, 41 us at 512 frames whatever the tile or the depth).  The
// weights still come straight from global memory (each element is used once per workgroup), kDepth chunks ahead; the K range (tap,
// 32-channel chunk) is split over the four waves as above, one reduction through LDS (over the consumed images), same epilogue.
template <int kDepth>
__global__ __launch_bounds__(256, 1) void f8_lds_kernel(const IgemmParams p, const int n_chunks, const int g_img, const int pitch) {
  using T = bf16raw;
  constexpr int TM = 64, TN = 64, FM = 4, FN = 4, NW = 4;
  constexpr int PITCH = TN + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char f8_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int img0 = (int)blockIdx.x * g_img;
  const int n_here = min(g_img, p.n_img - img0);
  const int rows_here = n_here * p.howo;      // valid output pixels of this workgroup
  const int hw_in = p.hi * p.wi;

  // ---- stage the images: pixel q (0 .. n_here * hi * wi) -> LDS offset q * pitch, 2 ci bytes = ci / 32 pieces of 64 lanes x 16 B...
  // one wave-instruction moves 1024 bytes: a pixel of ci = 512 channels is exactly one piece
  {
    const unsigned long long a_ptr = (unsigned long long)p.in + (unsigned long long)img0 * hw_in * p.pix_stride * 2;
    const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                        __builtin_amdgcn_readfirstlane(n_here * hw_in * p.pix_stride * 2), 0x00020000};
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)f8_smem;
    const int pieces_per_pix = p.ci * 2 / 1024;
    const int n_pieces = n_here * hw_in * pieces_per_pix;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int pc = wave_u; pc < n_pieces; pc += NW) {
      const int pix = pc / pieces_per_pix, part = pc - pix * pieces_per_pix;
      const uint32_t voff = (uint32_t)(pix * p.pix_stride * 2 + part * 1024 + lane * 16);
      const uint32_t l_ = lds_base + (uint32_t)(pix * pitch + part * 1024);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(l_), "s"(rs_a) : "memory");
    }
  }

  // per-lane rows: local output pixel r -> (image, oy, ox) -> LDS byte offset of input pixel (oy, ox) of that image
  uint32_t a_lds[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int r = i * 16 + i16;
    r = r < rows_here ? r : rows_here - 1;
    const int im = r / p.howo, rem = r - im * p.howo;
    const int oy = rem / p.wo, ox = rem - oy * p.wo;
    a_lds[i] = (uint32_t)((im * hw_in + oy * p.wi + ox) * pitch + g * 16);
  }
  const unsigned char* __restrict__ w_base[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) w_base[j] = reinterpret_cast<const unsigned char*>(p.wt) + (long)(j * 16 + i16) * p.ktot * 2 + g * 16;

  const int my_chunks = (n_chunks - wave + NW - 1) / NW;
  const int cpt = p.ci / 32;  // 32-deep chunks per tap
  uint4 fb[kDepth][FN];
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_w = [&](int d, int c) {
    c = c < my_chunks ? c : (my_chunks > 0 ? my_chunks - 1 : 0);
    const long wo = (long)(wave + NW * c) * 64;
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[d][j] = *reinterpret_cast<const uint4*>(w_base[j] + wo);
  };
  if (my_chunks > 0) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) load_w(d, d);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (also the first weight fragments: simple, once per workgroup)
  __syncthreads();                                   // the images are in LDS for every wave
  if (my_chunks > 0) {
    for (int c = 0; c < my_chunks; c += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        if (c + d < my_chunks) {
          const int ch = wave + NW * (c + d);
          const int tap = ch / cpt, c32 = ch - tap * cpt;
          const int ky = tap / p.s, kx = tap - ky * p.s;
          const uint32_t toff = (uint32_t)((ky * p.wi + kx) * pitch + c32 * 64);
          uint4 fa[FM];
#pragma unroll
          for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const uint4*>(f8_smem + a_lds[i] + toff);
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[d][j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
          load_w(d, c + d + kDepth);
        }
      }
    }
  }
  __syncthreads();  // every wave is done reading the images: LDS becomes the reduction buffer
  float (*sRed)[TM][PITCH] = reinterpret_cast<float (*)[TM][PITCH]>(f8_smem);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(&sRed[wave][i * 16 + i16][j * 16 + g * 4]) = acc[i][j];
  __syncthreads();
  constexpr int TPR = TN / 4, RPP = 256 / TPR;
  const int er = tid / TPR, ec = (tid % TPR) * 4;
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  float bv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bv[e] = (p.bias && ec + e < p.co) ? p.bias[ec + e] : 0.f;
#pragma unroll
  for (int it = 0; it < TM / RPP; ++it) {
    const int r = it * RPP + er;
    if (r >= rows_here || ec >= p.co) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(&sRed[0][r][ec]);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&sRed[w][r][ec]);
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    T ov[4];
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[e] = v[e] + bv[e];
      if (p.relu) x[e] = fmaxf(x[e], 0.f);
      ov[e] = from_f32<T>(x[e]);
    }
    const size_t m = (size_t)img0 * p.howo + r;
    if (out) *reinterpret_cast<uint2*>(out + m * p.ldo + ec) = *reinterpret_cast<const uint2*>(ov);
    if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + m * p.ldo + ec) = make_float4(x[0], x[1], x[2], x[3]);
  }
}

}  // namespace

// rows of `out` one entry of colsum_part covers when vdqn_launch_skinny takes a call (the generic kernels: 128)
int vdqn_skinny_part_rows(int conv) { return conv ? 64 : 32; }  // (the convolution variant is never asked for column sums)

bool vdqn_skinny_enabled() {
  static const bool on = [] { const char* e = getenv("VDQN_SKINNY"); return !(e && e[0] == '0'); }();
  return on;
}

// Does vdqn_conv2d hand this call to the skinny kernels?  bf16, a linear layer (1x1 over 1x1 "images": M = samples) or a valid
// (pad 0) stride-1 r x s convolution with at most 64 output columns, forward or (linear only) data gradient, M small enough that the
// generic kernel would be a handful of tiles; no sibling, no residual.
int vdqn_skinny_kind(const vdqn_conv_args* a) {
  if (!vdqn_skinny_enabled() || a->dtype != VDQN_BF16 || a->wt2 || a->in2 || a->resid) return 0;
  if (a->ci % 32 != 0 || (a->pix_stride * 2) % 16 != 0 || a->ldo % 4 != 0 || a->co % 4 != 0) return 0;
  if ((((uintptr_t)a->out | (uintptr_t)a->mask) & 7) != 0 || (((uintptr_t)a->out_f32) & 15) != 0) return 0;
  const long long M = (long long)a->n_img * a->ho * a->wo;
  const bool linear = a->r == 1 && a->s == 1 && a->hi == 1 && a->wi == 1 && a->ho == 1 && a->wo == 1 && a->stride == 1 && a->pad == 0;
  if (linear && M <= 4096 && a->co % 32 == 0) return 1;
  const bool valid_conv = a->mode == 0 && a->stride == 1 && a->pad == 0 && a->r * a->s > 1 && a->ho == a->hi - a->r + 1 && a->wo == a->wi - a->s + 1 &&
                          a->pix_stride == a->ci && !a->mask && !a->colsum_part;
  if (valid_conv && a->co == 64 && M <= 65536) return 2;
  return 0;
}

extern "C" int32_t vdqn_conv2d_colsum_rows(const vdqn_conv_args* a) {
  if (!a) return 128;
  const int kind = vdqn_skinny_kind(a);
  return kind ? vdqn_skinny_part_rows(kind == 2) : 128;
}

namespace {
template <int TM, int TN, bool CONV, int D, int NW = 4>
void launch_cfg(IgemmParams& p, int n_chunks, hipStream_t stream) {
  p.tiles_m = (p.M + TM - 1) / TM;
  p.tiles_n = (p.co + TN - 1) / TN;
  constexpr size_t smem = (size_t)NW * TM * (TN + 4) * 4;
  vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&skinny_kernel<TM, TN, CONV, D, NW>), smem);
  hipLaunchKernelGGL((skinny_kernel<TM, TN, CONV, D, NW>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(64 * NW), smem, stream, p, n_chunks,
                     make_fastdiv((uint32_t)(CONV ? p.wo : 1)), make_fastdiv((uint32_t)(CONV ? p.howo : 1)));
}
}  // namespace

int vdqn_launch_skinny(const void* pv, int kind, hipStream_t stream) {
  IgemmParams p = *reinterpret_cast<const IgemmParams*>(pv);
  const int n_chunks = p.ktot / 32;
  const double flops = 2.0 * p.M * p.co * p.ktot;
  const double bytes = 2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.mask != nullptr)));
  if (kind == 1) {
    vdqn_prof_begin(p.mask ? "skinny<bf16,32x32,dgrad>" : "skinny<bf16,32x32,fwd>", flops, bytes, stream);
    launch_cfg<32, 32, false, 4>(p, n_chunks, stream);
  } else {
    // VDQN_SKINNY_CONV_CFG: tile / prefetch-depth variants of the features.8 kernel (measurement switch)
    static const int cfg = [] { const char* e = getenv("VDQN_SKINNY_CONV_CFG"); return e ? atoi(e) : 0; }();
    vdqn_prof_begin("skinny<bf16,conv>", flops, bytes, stream);
    // default (VDQN_SKINNY_CONV_CFG unset or 8): the input images of a workgroup in LDS (f8_lds_kernel) where the geometry allows it:
    // pixels of whole 1 KB pieces, at least one image per 64-row tile, the images of a tile within 150 KB
    {
      const int g_img = p.howo > 0 ? 64 / p.howo : 0;
      const int pitch = p.ci * 2 + 32;  // +8 banks per pixel, +4 per K group: the 16 lanes of a ds_read_b128 group cover all 64 banks once
      const size_t img_bytes = (size_t)g_img * p.hi * p.wi * pitch;
      const size_t smem = img_bytes > (size_t)4 * 64 * 68 * 4 ? img_bytes : (size_t)4 * 64 * 68 * 4;
      if ((cfg == 0 || cfg == 8) && g_img >= 1 && (p.ci * 2) % 1024 == 0 && p.pix_stride == p.ci && smem <= 150 * 1024 && p.co == 64 &&
          (long long)p.n_img * p.hi * p.wi * p.ci * 2 < 0x7fffffffLL) {
        // (four weight chunks in flight per wave: 26.3 us at 512 frames; eight 27.9, twelve 29.6 — profiles/r04r_bench_head_f8_lds.txt)
        vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&f8_lds_kernel<4>), smem);
        hipLaunchKernelGGL((f8_lds_kernel<4>), dim3((unsigned)((p.n_img + g_img - 1) / g_img)), dim3(256), smem, stream, p, n_chunks, g_img, pitch);
        vdqn_prof_end(stream);
        VDQN_LAUNCH_CHECK();
        return VDQN_OK;
      }
    }
    // default: 64 x 64 tiles; a launch of at most 128 of them (the target network's 256 frames) runs on 32-row tiles instead — twice
    // the workgroups on a chip that would be half empty (measured 39 -> 32 us at 256 frames, 41 -> 60 us at 512: tools/bench_head.py)
    switch ((cfg && cfg != 9) ? cfg : (p.M <= 64 * 128 ? 1 : 0)) {  // (9: the global-memory kernel with its default tile choice)
      case 1: launch_cfg<32, 64, true, 6>(p, n_chunks, stream); break;
      case 2: launch_cfg<32, 64, true, 8>(p, n_chunks, stream); break;
      case 3: launch_cfg<64, 32, true, 6>(p, n_chunks, stream); break;
      case 4: launch_cfg<32, 32, true, 8>(p, n_chunks, stream); break;
      case 5: launch_cfg<64, 64, true, 2>(p, n_chunks, stream); break;
      // eight waves (twice the loads in flight per CU): 43.9 / 42.2 us at 512 / 256 frames against 40.8 / 31.7 — the kernel is not
      // bound by loads in flight; its im2col re-reads (9 taps of a 125 KB working set per workgroup) come from beyond L2
      // (profiles/r04p_bench_head_f8_eight_waves.txt)
      case 6: launch_cfg<64, 64, true, 4, 8>(p, n_chunks, stream); break;
      case 7: launch_cfg<32, 64, true, 6, 8>(p, n_chunks, stream); break;
      default: launch_cfg<64, 64, true, 4>(p, n_chunks, stream); break;
    }
  }
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
