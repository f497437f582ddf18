// Streaming kernel for the 1x1 / stride-2 `downsample` convolutions of ResNet layer2.0 / layer3.0 / layer4.0, forward, bf16
// (torchvision resnet.py BasicBlock.downsample, reached from archs/HabitatDQNMultiAction.py:30,49-51): out[m][:] =
// W[2C][C] x[pixel(2 oy, 2 ox)][:] + bias (BatchNorm folded), no ReLU — the shortcut operand of the block's second convolution.
//
// These three layers are HBM-bound (21-85 FLOP per byte: one input pixel in four is read, an output twice as wide is written) with
// K = 64 / 128 / 256, i.e. ONE, two or four K-steps of the tiled implicit GEMM, whose per-tile prologue (descriptor set-up, the
// first DMA round trip, a barrier) and epilogue then cost more than the matrix work: 135 TFLOP/s = 0.054 of peak, 0.13 ms per update.
// So this kernel has no tiles, no LDS and no barrier:
//   * the WEIGHTS LIVE IN REGISTERS for the life of a wave (the MFMA's first operand; C = 64: all 128 x 64 of them in 64 VGPRs;
//     C = 128 / 256: the N range is split over 2 / 8 waves, 128 VGPRs each), loaded once per wave;
//   * every wave streams 16-row fragments of input pixels straight from global memory into MFMA fragments (a pixel's C channels
//     are K-contiguous: one 16-byte load per lane and 32-deep chunk), MF fragments in flight, grid-stride over the rows;
//   * the weight rows are fetched in a permuted order so that a lane ends with 16 / 32 CONSECUTIVE output channels of its pixel:
//     bias add and 16-byte stores straight from the accumulators, whole 128-byte lines per row.
// Same products, one f32 accumulation chain over K in ascending chunks of 32, bias added to the finished sum: the arithmetic of the
// tiled kernel.
#include <stdlib.h>

#include <type_traits>

#include "igemm_common.h"

namespace {

// NW = waves that share one row range and split the 2C output channels; RG = row groups per workgroup (RG * NW waves)
template <int C, int NW, int RG, int MF>
__global__ __launch_bounds__(64 * NW * RG, 1) void ds1x1_kernel(const IgemmParams p, const int n_frags, const FastDiv d_wo, const FastDiv d_howo) {
  using T = bf16raw;
  constexpr int CO = 2 * C;
  constexpr int NFW = CO / NW / 16;  // output fragments per wave
  constexpr int KC = C / 32;         // 32-deep chunks
  constexpr int CPL = 4 * NFW;       // consecutive output channels per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int nw = wave % NW, rg = wave / NW;
  const int n_base = nw * (CO / NW);

  // weights: fragment j of this wave, MFMA row i16 = 4 g' + r'  <->  output channel n_base + g' * CPL + 4 j + r'
  uint4 wf[NFW][KC];
  {
    const int ch_lane = n_base + (i16 >> 2) * CPL + (i16 & 3);
#pragma unroll
    for (int j = 0; j < NFW; ++j)
#pragma unroll
      for (int k = 0; k < KC; ++k)
        wf[j][k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.wt) + (size_t)(ch_lane + 4 * j) * C + k * 32 + g * 8);
  }
  float bv[CPL];
#pragma unroll
  for (int e = 0; e < CPL; ++e) bv[e] = p.bias ? p.bias[n_base + g * CPL + e] : 0.f;

  const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const int groups = gridDim.x * RG;               // row groups in the grid
  const int my_group = blockIdx.x * RG + rg;
  // a row group walks blocks of MF fragments (block b = fragments [b * MF, b * MF + MF)), the NEXT block's pixels loaded into the
  // second register set before the current block is multiplied and stored: the loads of two blocks are always in flight
  const int n_blocks = (n_frags + MF - 1) / MF;
  uint4 fa[2][MF][KC];
  int mrow[2][MF];
  auto load = [&](auto SET, int blk) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      int m = (blk * MF + f) * 16 + i16;
      mrow[S][f] = m;
      m = m < p.M ? m : p.M - 1;
      const uint32_t img = fastdiv((uint32_t)m, d_howo);
      const uint32_t rem = (uint32_t)m - img * (uint32_t)p.howo;
      const uint32_t oy = fastdiv(rem, d_wo);
      const uint32_t ox = rem - oy * (uint32_t)p.wo;
      const size_t pix = ((size_t)img * p.hi + 2 * oy) * p.wi + 2 * ox;
      const T* src = in + pix * C + g * 8;
#pragma unroll
      for (int k = 0; k < KC; ++k) fa[S][f][k] = *reinterpret_cast<const uint4*>(src + k * 32);
    }
  };
  auto compute = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      f32x4 acc[NFW];
#pragma unroll
      for (int j = 0; j < NFW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int j = 0; j < NFW; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j][k]), __builtin_bit_cast(bf16x8, fa[S][f][k]), acc[j], 0, 0, 0);
      if (mrow[S][f] < p.M) {
        T* dst = out + (size_t)mrow[S][f] * p.ldo + n_base + g * CPL;
#pragma unroll
        for (int q = 0; q < CPL / 8; ++q) {
          T ov[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int c = q * 8 + e;  // channel g * CPL + c = fragment c / 4, register c % 4
            ov[e] = from_f32<T>(acc[c >> 2][c & 3] + bv[c]);
          }
          reinterpret_cast<uint4*>(dst)[q] = *reinterpret_cast<const uint4*>(ov);
        }
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  int blk = my_group;
  if (blk < n_blocks) load(S0{}, blk);
  while (blk < n_blocks) {
    int nb = blk + groups;
    if (nb < n_blocks) load(S1{}, nb);
    compute(S0{});
    blk = nb;
    if (blk >= n_blocks) break;
    nb = blk + groups;
    if (nb < n_blocks) load(S0{}, nb);
    compute(S1{});
    blk = nb;
  }
}

template <int C, int NW, int RG, int MF>
void launch_ds(const IgemmParams& p, hipStream_t stream) {
  const int n_frags = (p.M + 15) / 16;
  const int n_blocks = (n_frags + MF - 1) / MF;
  const int wg_needed = (n_blocks + RG - 1) / RG;
  // two waves per SIMD fit (166-220 VGPRs): 8 waves per CU, each with MF x KC 16-byte loads per lane in flight (64-128 KB per CU);
  // grid-stride beyond that
  const int cap = vdqn_num_cus() * (8 / (NW * RG));
  const int grid = wg_needed < cap ? wg_needed : cap;
  hipLaunchKernelGGL((ds1x1_kernel<C, NW, RG, MF>), dim3((unsigned)grid), dim3(64 * NW * RG), 0, stream, p, n_frags, make_fastdiv((uint32_t)p.wo),
                     make_fastdiv((uint32_t)p.howo));
}

}  // namespace

// Does vdqn_conv2d hand this call to the streaming kernel?  (VDQN_DS_STREAM=0 keeps the tiled kernel.)
bool vdqn_ds1x1_takes(const vdqn_conv_args* a) {
  static const bool on = [] { const char* e = getenv("VDQN_DS_STREAM"); return !(e && e[0] == '0'); }();
  if (!on || a->dtype != VDQN_BF16 || a->mode != 0 || a->r != 1 || a->s != 1 || a->stride != 2 || a->pad != 0) return false;
  if (a->wt2 || a->in2 || a->wt_b || a->resid || a->mask || a->relu || a->out_f32 || a->colsum_part || !a->out) return false;
  if (!(a->ci == 64 || a->ci == 128 || a->ci == 256) || a->co != 2 * a->ci || a->pix_stride != a->ci || a->ldo % 8 != 0) return false;
  if (a->hi != 2 * a->ho || a->wi != 2 * a->wo) return false;
  if ((((uintptr_t)a->in | (uintptr_t)a->wt | (uintptr_t)a->out) & 15) != 0) return false;
  return (long long)a->n_img * a->ho * a->wo < (1ll << 24);
}

int vdqn_launch_ds1x1(const void* pv, hipStream_t stream) {
  const IgemmParams& p = *reinterpret_cast<const IgemmParams*>(pv);
  vdqn_prof_begin("ds1x1<bf16,fwd>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.M * p.ci + (double)p.co * p.ktot + (double)p.M * p.co), stream);
  if (p.ci == 64) launch_ds<64, 1, 4, 4>(p, stream);
  else if (p.ci == 128) launch_ds<128, 2, 2, 1>(p, stream);
  else launch_ds<256, 8, 1, 1>(p, stream);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
