// Implicit-GEMM convolution / linear kernel for gfx950 (forward and data-gradient).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),   m = (img, ho, wo),  k = (r, s, c)
//
// * A is gathered on the fly from the NHWC activation tensor: one K-step is 128 contiguous bytes of one
//   (r, s) tap of one input pixel (64 bf16 / 32 f32 channels), fetched as eight 16-byte lanes per row so
//   every global request is a full 128-byte line.  Padding taps are zero-filled in registers.
// * Tiles: 128 (M) x BN (N, 64 or 128) x 128 bytes (K); 256 threads = 4 waves in a 2x2 grid, each wave owns
//   64 x BN/2 as 16x16 MFMA fragments (v_mfma_f32_16x16x32_bf16, or v_mfma_f32_16x16x4_f32 in the exact
//   f32 parity mode — same staging code, only the MFMA differs).
// * LDS: rows of 128 B, the 16-byte chunk index XOR-swizzled with (row & 7) so the ds_read_b128 fragment
//   reads are bank-conflict free; two buffers, register-staged prefetch (global loads of step k+1 are in
//   flight while step k's MFMAs run), one barrier per K-step.
// * blockIdx is remapped so the N-tiles of one M-tile run on the same XCD (A rows stay in that XCD's L2).
//
// Reference call sites this serves: the torch conv2d/linear (+BatchNorm eval, ReLU, residual) launched
// from archs/HabitatDQNMultiAction.py:30-31,49-53 and their backward (train_q_network.py:226).
#include "common.h"

namespace {

struct IgemmParams {
  const void* in;
  const void* wt;
  const float* bias;
  const void* resid;
  const void* mask;
  void* out;
  float* out_f32;
  int n_img, hi, wi, ci, pix_stride, ho, wo, co, ldo, r, s, stride, pad, mode, relu;
  int M, howo, ktot, nk, tiles_m, tiles_n;
};

template <typename T, int BN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmParams p) {
  constexpr int BM = 128;
  constexpr int E16 = 16 / (int)sizeof(T);
  constexpr int KC = 128 / (int)sizeof(T);
  constexpr int NF = BN / 32;
  constexpr int BROWS = BN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * BM * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lb % (uint32_t)p.tiles_n), tile_m = (int)(lb / (uint32_t)p.tiles_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int lchunk = tid & 7, lrow = tid >> 3;

  const T* __restrict__ in = (const T*)p.in;
  const T* __restrict__ wt = (const T*)p.wt;

  // ---- per-row gather bases (4 A rows per thread) ----
  const T* a_ptr[4];
  int a_h0[4], a_w0[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lrow + 32 * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int img = mm / p.howo;
    const int rem = mm - img * p.howo;
    const int oh = rem / p.wo;
    const int ow = rem - oh * p.wo;
    a_ok[i] = ok;
    a_ptr[i] = in + (size_t)img * p.hi * p.wi * p.pix_stride + lchunk * E16;
    if (p.mode == 0) {
      a_h0[i] = oh * p.stride - p.pad;
      a_w0[i] = ow * p.stride - p.pad;
    } else {
      a_h0[i] = oh + p.pad;
      a_w0[i] = ow + p.pad;
    }
  }
  const T* b_ptr[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_ptr[i] = wt + (size_t)(n0 + lrow + 32 * i) * p.ktot + lchunk * E16;

  uint4 ra[4], rb[BROWS];
  auto load_tile = [&](int kr, int ks, int c0, int kstep) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int h, w;
      bool ok = a_ok[i];
      if (p.mode == 0) {
        h = a_h0[i] + kr;
        w = a_w0[i] + ks;
      } else {
        const int th = a_h0[i] - kr, tw = a_w0[i] - ks;
        if (p.stride == 2) {
          ok = ok && (((th | tw) & 1) == 0);
          h = th >> 1;
          w = tw >> 1;
        } else {
          h = th;
          w = tw;
        }
      }
      ok = ok && ((unsigned)h < (unsigned)p.hi) && ((unsigned)w < (unsigned)p.wi);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = *reinterpret_cast<const uint4*>(a_ptr[i] + ((h * p.wi + w) * p.pix_stride + c0));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) rb[i] = *reinterpret_cast<const uint4*>(b_ptr[i] + kstep * KC);
  };
  const int swz_w = ((lchunk ^ (lrow & 7)) << 4);
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(sA + buf * (BM * 128) + (lrow + 32 * i) * 128 + swz_w) = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      *reinterpret_cast<uint4*>(sB + buf * (BN * 128) + (lrow + 32 * i) * 128 + swz_w) = rb[i];
  };

  f32x4 acc[4][NF];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, g = lane >> 4;
  const int sw = i16 & 7;
  auto compute = [&](int buf) {
    const unsigned char* a = sA + buf * (BM * 128) + (wr * 64 + i16) * 128;
    const unsigned char* b = sB + buf * (BN * 128) + (wc * (BN / 2) + i16) * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int coff = (((g + 4 * h) ^ sw) << 4);
      uint4 af[4], bfr[NF];
#pragma unroll
      for (int f = 0; f < 4; ++f) af[f] = *reinterpret_cast<const uint4*>(a + f * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < NF; ++j) bfr[j] = *reinterpret_cast<const uint4*>(b + j * 16 * 128 + coff);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[f]),
                                                                __builtin_bit_cast(bf16x8, bfr[j]), acc[f][j], 0, 0, 0);
          } else {
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].x), __uint_as_float(bfr[j].x), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].y), __uint_as_float(bfr[j].y), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].z), __uint_as_float(bfr[j].z), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].w), __uint_as_float(bfr[j].w), acc[f][j], 0, 0, 0);
          }
        }
    }
  };

  // ---- main loop: K-steps enumerate (r, s, c0) with c0 fastest ----
  int kr = 0, ks = 0, c0 = 0;
  load_tile(kr, ks, c0, 0);
  for (int k = 0; k < p.nk; ++k) {
    const int buf = k & 1;
    store_tile(buf);
    __syncthreads();
    if (k + 1 < p.nk) {
      c0 += KC;
      if (c0 >= p.ci) {
        c0 = 0;
        if (++ks == p.s) {
          ks = 0;
          ++kr;
        }
      }
      load_tile(kr, ks, c0, k + 1);
    }
    compute(buf);
  }

  // ---- epilogue: C layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg ----
  T* __restrict__ out = (T*)p.out;
  const T* __restrict__ resid = (const T*)p.resid;
  const T* __restrict__ mask = (const T*)p.mask;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int n = n0 + wc * (BN / 2) + j * 16 + i16;
      if (n >= p.co) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + wr * 64 + f * 16 + g * 4 + reg;
        if (m >= p.M) continue;
        const size_t o = (size_t)m * p.ldo + n;
        float v = acc[f][j][reg] + bv;
        if (resid) v += to_f32<T>(resid[o]);
        if (p.relu) v = fmaxf(v, 0.f);
        if (mask) v = (to_f32<T>(mask[o]) > 0.f) ? v : 0.f;
        if (out) out[o] = from_f32<T>(v);
        if (p.out_f32) p.out_f32[o] = v;
      }
    }
  }
}

template <typename T, int BN>
int launch_igemm(const IgemmParams& p, hipStream_t stream) {
  const size_t smem = 2 * (128 + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
  const double esz = sizeof(T);
  vdqn_prof_begin(sizeof(T) == 2 ? (BN == 128 ? "igemm<bf16,128>" : "igemm<bf16,64>") : (BN == 128 ? "igemm<f32,128>" : "igemm<f32,64>"),
                  2.0 * p.M * p.co * p.ktot,
                  esz * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))),
                  stream);
  hipLaunchKernelGGL((igemm_kernel<T, BN>), dim3(grid), dim3(256), smem, stream, p);
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

}  // namespace

extern "C" int vdqn_conv2d(const vdqn_conv_args* a, void* stream) {
  VDQN_CHECK(a != nullptr, "vdqn_conv2d: null args");
  VDQN_CHECK(a->dtype == VDQN_F32 || a->dtype == VDQN_BF16, "vdqn_conv2d: bad dtype %d", a->dtype);
  const int kc = a->dtype == VDQN_BF16 ? 64 : 32;
  VDQN_CHECK(a->in && a->wt && (a->out || a->out_f32), "vdqn_conv2d: null tensor");
  VDQN_CHECK(a->ci > 0 && a->ci % kc == 0, "vdqn_conv2d: ci=%d must be a multiple of %d", a->ci, kc);
  VDQN_CHECK(a->stride == 1 || a->stride == 2, "vdqn_conv2d: stride %d unsupported", a->stride);
  VDQN_CHECK(a->mode == 0 || a->mode == 1, "vdqn_conv2d: bad mode %d", a->mode);
  VDQN_CHECK(a->n_img > 0 && a->ho > 0 && a->wo > 0 && a->co > 0 && a->ldo >= a->co, "vdqn_conv2d: bad dims");
  VDQN_CHECK((int64_t)a->n_img * a->ho * a->wo < (1ll << 31) && (int64_t)a->hi * a->wi * a->pix_stride < (1ll << 31),
             "vdqn_conv2d: tensor too large");
  IgemmParams p;
  p.in = a->in; p.wt = a->wt; p.bias = a->bias; p.resid = a->resid; p.mask = a->mask; p.out = a->out; p.out_f32 = a->out_f32;
  p.n_img = a->n_img; p.hi = a->hi; p.wi = a->wi; p.ci = a->ci; p.pix_stride = a->pix_stride;
  p.ho = a->ho; p.wo = a->wo; p.co = a->co; p.ldo = a->ldo; p.r = a->r; p.s = a->s; p.stride = a->stride; p.pad = a->pad;
  p.mode = a->mode; p.relu = a->relu;
  p.M = a->n_img * a->ho * a->wo;
  p.howo = a->ho * a->wo;
  p.ktot = a->r * a->s * a->ci;
  p.nk = p.ktot / kc;
  const int bn = (a->co % 128 == 0) ? 128 : 64;
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (a->co + bn - 1) / bn;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == VDQN_BF16) return bn == 128 ? launch_igemm<bf16raw, 128>(p, st) : launch_igemm<bf16raw, 64>(p, st);
  return bn == 128 ? launch_igemm<float, 128>(p, st) : launch_igemm<float, 64>(p, st);
}
