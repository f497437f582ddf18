// Weight-gradient implicit GEMM for gfx950:
//
//   dw[n][r][s][c] += sum_m gy[m, n] * x[pix(m, r, s), c]        (m = output pixel, the reduction index)
//
// Both operands are stored pixel-major (channels contiguous), i.e. transposed w.r.t. what an MFMA A/B
// fragment wants (k = pixel contiguous per lane).  The tiles are therefore staged into LDS exactly as they
// lie in HBM ([32 pixels][BT channels], full-line 16-byte loads) and the transpose is done by the LDS read:
// ds_read_b64_tr_b16 for bf16 (4 pixels x 16 channels per 16-lane group, delivered channel-per-lane), plain
// ds_read_b32 for the f32 16x16x4 MFMA whose fragment is one element per lane.
//
// Grid: x = (co tile, tap, ci tile), y = split of the pixel range; partial tiles are summed with
// global_atomic_add_f32 into the pre-zeroed f32 dw (all lanes of a row write 64 contiguous bytes).
// LDS rows are XOR-swizzled on the 16-byte chunk index so the transposing reads of a half-wave (8 pixel rows
// x 32 B) fall on distinct banks.
//
// Replaces the convolution_backward / addmm weight-gradient kernels behind loss.backward()
// (train_q_network.py:226).
#include "common.h"

namespace {

struct WgradParams {
  const void* gy;
  const void* x;
  float* dw;
  int n_img, hi, wi, ci, pix_stride, ho, wo, co, ldg, r, s, stride, pad;
  int M, kchunk, taps, ci_tiles;
  FastDiv d_howo, d_wo;
};

template <typename T, int BT> __device__ __forceinline__ int wg_swz(int row) {
  if constexpr (sizeof(T) == 2) {
    if constexpr (BT == 128) return (row & 7) << 1;
    else return ((row >> 1) & 3) << 1;
  } else {
    return (row & 1) << 2;
  }
}

template <typename T, int BT>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  constexpr int E16 = 16 / (int)sizeof(T);
  constexpr int RB = BT * (int)sizeof(T);  // bytes per LDS row (one pixel)
  constexpr int CPR = RB / 16;             // 16-byte chunks per row
  constexpr int NL = (32 * CPR) / 256;     // chunks per thread per operand per K-step
  constexpr int NFR = BT / 32;             // 16-wide fragments per wave per dim
  constexpr int KP = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                 // [2][KP*RB]
  unsigned char* sB = smem + 2 * KP * RB;   // [2][KP*RB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int ci_tile = t % p.ci_tiles;
  t /= p.ci_tiles;
  const int tap = t % p.taps;
  const int co_tile = t / p.taps;
  const int kr = tap / p.s, ks = tap - kr * p.s;
  const int co0 = co_tile * BT, ci0 = ci_tile * BT;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.M, kbeg + p.kchunk);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + KP - 1) / KP;

  const T* __restrict__ gy = (const T*)p.gy;
  const T* __restrict__ x = (const T*)p.x;

  int l_row[NL], l_ch[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int q = tid + 256 * i;
    l_row[i] = q / CPR;
    l_ch[i] = q % CPR;
  }

  uint4 ra[NL], rb[NL];
  auto load_tile = [&](int kb) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int pm = kb + l_row[i];
      const bool ok = pm < kend;
      uint4 va = make_uint4(0u, 0u, 0u, 0u), vb = make_uint4(0u, 0u, 0u, 0u);
      if (ok) {
        va = *reinterpret_cast<const uint4*>(gy + (size_t)pm * p.ldg + co0 + l_ch[i] * E16);
        const uint32_t img = fastdiv((uint32_t)pm, p.d_howo);
        const uint32_t rem = (uint32_t)pm - img * p.d_howo.div;
        const uint32_t oh = fastdiv(rem, p.d_wo);
        const uint32_t ow = rem - oh * p.d_wo.div;
        const int h = (int)oh * p.stride - p.pad + kr;
        const int w = (int)ow * p.stride - p.pad + ks;
        if ((unsigned)h < (unsigned)p.hi && (unsigned)w < (unsigned)p.wi)
          vb = *reinterpret_cast<const uint4*>(x + ((size_t)img * p.hi * p.wi + (size_t)(h * p.wi + w)) * p.pix_stride + ci0 +
                                               l_ch[i] * E16);
      }
      ra[i] = va;
      rb[i] = vb;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int off = l_row[i] * RB + ((l_ch[i] ^ wg_swz<T, BT>(l_row[i])) << 4);
      *reinterpret_cast<uint4*>(sA + buf * (KP * RB) + off) = ra[i];
      *reinterpret_cast<uint4*>(sB + buf * (KP * RB) + off) = rb[i];
    }
  };

  f32x4 acc[NFR][NFR];
#pragma unroll
  for (int f = 0; f < NFR; ++f)
#pragma unroll
    for (int j = 0; j < NFR; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave >> 1, wc = wave & 1;
  const int grp = lane >> 4, i16 = lane & 15;

  auto compute = [&](int buf) {
    const unsigned char* a = sA + buf * (KP * RB);
    const unsigned char* b = sB + buf * (KP * RB);
    if constexpr (sizeof(T) == 2) {
      // transposing read: lane 4q+pp of a 16-lane group addresses pixel row (4*grp + q), channels 4pp..4pp+3
      const int q = i16 >> 2, pp = i16 & 3;
      const int row = 4 * grp + q;
      const int sz = wg_swz<T, BT>(row);
      const int sub = (pp & 1) << 3;
      s16x8 af[NFR], bfr[NFR];
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        const int cb = wr * (BT / 2) + f * 16;
        const int off = row * RB + ((((cb >> 3) + (pp >> 1)) ^ sz) << 4) + sub;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + off + 16 * RB));
        af[f] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const int cb = wc * (BT / 2) + j * 16;
        const int off = row * RB + ((((cb >> 3) + (pp >> 1)) ^ sz) << 4) + sub;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + off + 16 * RB));
        bfr[j] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int f = 0; f < NFR; ++f)
#pragma unroll
        for (int j = 0; j < NFR; ++j)
          acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[f]), __builtin_bit_cast(bf16x8, bfr[j]),
                                                              acc[f][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int row = 4 * st + grp;
        const int sz = wg_swz<T, BT>(row);
        float av[NFR], bv[NFR];
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
          const int col = wr * (BT / 2) + f * 16 + i16;
          av[f] = *reinterpret_cast<const float*>(a + row * RB + (((col >> 2) ^ sz) << 4) + ((col & 3) << 2));
        }
#pragma unroll
        for (int j = 0; j < NFR; ++j) {
          const int col = wc * (BT / 2) + j * 16 + i16;
          bv[j] = *reinterpret_cast<const float*>(b + row * RB + (((col >> 2) ^ sz) << 4) + ((col & 3) << 2));
        }
#pragma unroll
        for (int f = 0; f < NFR; ++f)
#pragma unroll
          for (int j = 0; j < NFR; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f], bv[j], acc[f][j], 0, 0, 0);
      }
    }
  };

  load_tile(kbeg);
  for (int k = 0; k < nk; ++k) {
    const int buf = k & 1;
    store_tile(buf);
    __syncthreads();
    if (k + 1 < nk) load_tile(kbeg + (k + 1) * KP);
    compute(buf);
  }

  // C layout: col (lane & 15) -> ci, row ((lane >> 4) * 4 + reg) -> co
  const size_t row_len = (size_t)p.taps * p.ci;
#pragma unroll
  for (int f = 0; f < NFR; ++f)
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
      const int ci = ci0 + wc * (BT / 2) + j * 16 + i16;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = co0 + wr * (BT / 2) + f * 16 + grp * 4 + reg;
        if (co < p.co) atomicAdd(p.dw + (size_t)co * row_len + (size_t)tap * p.ci + ci, acc[f][j][reg]);
      }
    }
}

// dbias[c] += sum_m gy[m][c]: 16-byte column groups x row stripes per block, stripes reduced through LDS,
// one atomic per column per block (<= 256 blocks).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ gy, float* __restrict__ dbias, int M, int C, int ldg,
                                                     int rows_per_block) {
  constexpr int E16 = 16 / (int)sizeof(T);
  __shared__ float red[256 * E16];
  const int cg_n = C / E16;            // 16-byte column groups (<= 256, power-of-two multiples of 8)
  const int nstripe = 256 / cg_n;      // row stripes per block
  const int cg = threadIdx.x % cg_n, stripe = threadIdx.x / cg_n;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s[E16];
#pragma unroll
  for (int e = 0; e < E16; ++e) s[e] = 0.f;
  if (stripe < nstripe) {
    for (int r = r0 + stripe; r < r1; r += nstripe) {
      const uint4 v = *reinterpret_cast<const uint4*>(gy + (size_t)r * ldg + cg * E16);
      const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int e = 0; e < E16; ++e) s[e] += to_f32<T>(pv[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < E16; ++e) red[threadIdx.x * E16 + e] = s[e];
  __syncthreads();
  // thread t < C sums column t over the stripes
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / E16, e = c % E16;
    float t = 0.f;
    for (int st = 0; st < nstripe; ++st) t += red[(st * cg_n + g) * E16 + e];
    atomicAdd(dbias + c, t);
  }
}

template <typename T, int BT>
int launch_wgrad(const WgradParams& p, int tiles, int splitk, hipStream_t stream) {
  const size_t smem = 4 * 32 * BT * sizeof(T);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const double esz = sizeof(T);
  vdqn_prof_begin(sizeof(T) == 2 ? (BT == 128 ? "wgrad<bf16,128>" : "wgrad<bf16,64>") : (BT == 128 ? "wgrad<f32,128>" : "wgrad<f32,64>"),
                  2.0 * p.M * p.co * p.taps * p.ci,
                  esz * ((double)p.M * p.ldg + (double)p.n_img * p.hi * p.wi * p.ci) + 4.0 * p.co * p.taps * p.ci, stream);
  hipLaunchKernelGGL((wgrad_kernel<T, BT>), dim3(tiles, splitk), dim3(256), smem, stream, p);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

}  // namespace

extern "C" int vdqn_conv2d_wgrad(const vdqn_wgrad_args* a, void* stream) {
  VDQN_CHECK(a != nullptr, "vdqn_conv2d_wgrad: null args");
  VDQN_CHECK(a->dtype == VDQN_F32 || a->dtype == VDQN_BF16, "vdqn_conv2d_wgrad: bad dtype %d", a->dtype);
  VDQN_CHECK(a->gy && a->x && a->dw, "vdqn_conv2d_wgrad: null tensor");
  VDQN_CHECK(a->ci % 64 == 0 && a->ldg % 64 == 0, "vdqn_conv2d_wgrad: ci=%d and ldg=%d must be multiples of 64", a->ci, a->ldg);
  VDQN_CHECK(a->stride == 1 || a->stride == 2, "vdqn_conv2d_wgrad: stride %d unsupported", a->stride);
  const int64_t M64 = (int64_t)a->n_img * a->ho * a->wo;
  VDQN_CHECK(M64 > 0 && M64 < (1 << 24), "vdqn_conv2d_wgrad: %lld output pixels out of range (< 2^24)", (long long)M64);
  VDQN_CHECK(a->ho * a->wo < 65536, "vdqn_conv2d_wgrad: ho*wo too large");
  const int co_pad = (a->co + 63) / 64 * 64;
  VDQN_CHECK(a->ldg >= co_pad, "vdqn_conv2d_wgrad: gy rows (ldg=%d) must hold co padded to 64 (%d)", a->ldg, co_pad);
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  p.gy = a->gy; p.x = a->x; p.dw = a->dw;
  p.n_img = a->n_img; p.hi = a->hi; p.wi = a->wi; p.ci = a->ci; p.pix_stride = a->pix_stride;
  p.ho = a->ho; p.wo = a->wo; p.co = a->co; p.ldg = a->ldg; p.r = a->r; p.s = a->s; p.stride = a->stride; p.pad = a->pad;
  p.M = (int)M64;
  p.taps = a->r * a->s;
  p.d_howo = make_fastdiv((uint32_t)(a->ho * a->wo));
  p.d_wo = make_fastdiv((uint32_t)a->wo);
  const int bt = (co_pad % 128 == 0 && a->ci % 128 == 0) ? 128 : 64;
  p.ci_tiles = a->ci / bt;
  const int tiles = (co_pad / bt) * p.taps * p.ci_tiles;
  int splitk = a->splitk;
  if (splitk <= 0) {
    splitk = (1024 + tiles - 1) / tiles;
    const int max_split = (p.M + 255) / 256;  // at least 8 K-steps per block
    if (splitk > max_split) splitk = max_split;
    if (splitk < 1) splitk = 1;
  }
  VDQN_CHECK(splitk <= 65535, "vdqn_conv2d_wgrad: splitk too large");
  p.kchunk = ((p.M + splitk - 1) / splitk + 31) / 32 * 32;
  int rc;
  if (a->dtype == VDQN_BF16) rc = bt == 128 ? launch_wgrad<bf16raw, 128>(p, tiles, splitk, st) : launch_wgrad<bf16raw, 64>(p, tiles, splitk, st);
  else rc = bt == 128 ? launch_wgrad<float, 128>(p, tiles, splitk, st) : launch_wgrad<float, 64>(p, tiles, splitk, st);
  if (rc != VDQN_OK) return rc;
  if (a->dbias) {
    const int e16 = a->dtype == VDQN_BF16 ? 8 : 4;
    VDQN_CHECK(co_pad / e16 <= 256, "vdqn_conv2d_wgrad: dbias path supports up to %d channels", 256 * e16);
    int blocks = (p.M + 511) / 512;
    if (blocks > 256) blocks = 256;
    const int rpb = (p.M + blocks - 1) / blocks;
    vdqn_prof_begin("colsum", 0.0, (double)p.M * co_pad * (a->dtype == VDQN_BF16 ? 2 : 4), st);
    if (a->dtype == VDQN_BF16)
      hipLaunchKernelGGL((colsum_kernel<bf16raw>), dim3(blocks), dim3(256), 0, st, (const bf16raw*)a->gy, a->dbias, p.M, co_pad, a->ldg, rpb);
    else
      hipLaunchKernelGGL((colsum_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)a->gy, a->dbias, p.M, co_pad, a->ldg, rpb);
    vdqn_prof_end(st);
    VDQN_LAUNCH_CHECK();
  }
  return VDQN_OK;
}
