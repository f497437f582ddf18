// Halo-tile implicit GEMM for the 64-output-channel, stride-1 layers (ResNet layer1's 3x3 convs, forward and
// data-gradient, and the space-to-depth stem).
//
// With only 64 output channels every staged activation byte feeds just 64 MACs, so the generic kernel
// (igemm.hip), which re-stages the im2col row for each of the 9 taps, is bound by the LDS fill rate (measured
// 6.7 TB/s of LDS-DMA at 250-400 TFLOP/s).  Here a block owns two 8x8 output sub-tiles; their input patches
// ((8+R-1) x (8+S-1) pixels, 12.5 KiB for 64 bf16 channels) go to LDS ONCE and every tap reads its A fragments
// from the patch at a shifted address — 5.6x less activation traffic into LDS.  Only the 8 KiB weight slice of
// the current tap is streamed (double-buffered LDS-DMA, one barrier per tap).
//
//   rows of the 128x64 tile:  row = sub*64 + py*8 + px   (sub-tile `sub`, pixel (py, px) of its 8x8 square)
//   A(row, k-step (kr,ks,cc)) = 128 bytes at patch[sub][(py+ky)*PW + px+kx][cc*128 ..], ky/kx = kr/ks (forward)
//                               or R-1-kr / S-1-ks (data gradient = correlation with the flipped taps)
//   stem: the 128-byte K row is 4 consecutive s2d pixels x 16 channels, so S = 1 and the patch is 3 pixels wider.
#include <stdlib.h>

#include "common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobH = 0x80000000u;

struct HaloParams {
  const void* in;
  const void* wt;
  const float* bias;
  const void* resid;
  const void* mask;
  void* out;
  float* out_f32;
  float* colsum_part;
  int n_img, hi, wi, pix_stride, ho, wo, co, ldo, R, S, pad, relu, dgrad;
  int span_px;   // pixels covered by one 128-byte-multiple K row (1, or 4 for the stem)
  int PW, PH;    // patch width / height in pixels
  int tiles_x, tiles_per_img, n_sub, ktot;
  long long in_bytes;
  int wt_bytes, vec_ok;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void igemm_halo_kernel(const HaloParams p) {
  constexpr int ESZ = (int)sizeof(T);
  constexpr int BN = 64, NF = 4, NSUB = 4;  // 4 sub-tiles per block, wave w owns sub-tile w with all 64 columns
  constexpr int LDC = BN + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int PB = p.pix_stride * ESZ;                 // bytes per input pixel
  const int patch_bytes = p.PH * p.PW * PB;
  const int NPA = (patch_bytes + 1023) >> 10;         // 1 KiB DMA pieces per patch
  const int patch_stride = NPA << 10;
  unsigned char* sA = smem;                           // [NSUB][patch_stride]
  unsigned char* sB = smem + NSUB * patch_stride;     // [2][64 * 128]
  const int cpt = (p.pix_stride * p.span_px * ESZ) >> 7;  // 128-byte chunks per tap
  const int nk = p.R * p.S * cpt;

  const int sub0 = blockIdx.x * NSUB;
  // ---- descriptors ----
  const int img0 = sub0 / p.tiles_per_img;
  const long long img_bytes = (long long)p.hi * p.wi * PB;
  const long long a_base_off = (long long)img0 * img_bytes;
  long long a_rem = p.in_bytes - a_base_off;
  if (a_rem > 0x7fffffffLL) a_rem = 0x7fffffffLL;
  const unsigned long long a_ptr = (unsigned long long)((const unsigned char*)p.in + a_base_off);
  const unsigned long long b_ptr = (unsigned long long)p.wt;
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)a_rem), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool swz = (PB % 128) == 0;  // 128-byte pixels: XOR the 16-byte chunk with (hx & 7); narrow pixels stay linear

  // ---- stage both input patches once: pieces wave, wave+4, ... of the 2*NPA ----
  for (int q = wave_u; q < NSUB * NPA; q += 4) {
    const int sub = q / NPA;
    const int piece = q - sub * NPA;
    const int st = sub0 + sub;
    const int o = piece * 1024 + lane * 16;           // byte offset inside the patch image
    const int hp = o / PB, pos = (o - hp * PB) >> 4;  // patch pixel, 16-byte slot inside the pixel
    const int hy = hp / p.PW, hx = hp - hy * p.PW;
    uint32_t vo = kOobH;
    if (st < p.n_sub && hp < p.PH * p.PW) {
      const int img = st / p.tiles_per_img;
      const int trem = st - img * p.tiles_per_img;
      const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
      const int y = ty * 8 - p.pad + hy, x = tx * 8 - p.pad + hx;
      if ((unsigned)y < (unsigned)p.hi && (unsigned)x < (unsigned)p.wi) {
        const int c = swz ? ((pos & ~7) | ((pos & 7) ^ (hx & 7))) : pos;
        vo = (uint32_t)(((img - img0) * p.hi + y) * p.wi + x) * (uint32_t)PB + (uint32_t)(c * 16);
      }
    }
    const uint32_t la = lds_base + (uint32_t)(sub * patch_stride + piece * 1024);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(vo), "s"(la), "s"(rs_a) : "memory");
  }
  // ---- weights: 64 rows x 128 B per K-step, 2 pieces per wave ----
  const int brow0 = wave * 8 + (lane >> 3), brow1 = brow0 + 32;
  const uint32_t b_off0 = (uint32_t)brow0 * (uint32_t)(p.ktot * ESZ) + (uint32_t)((((lane & 7) ^ (brow0 & 7))) * 16);
  const uint32_t b_off1 = (uint32_t)brow1 * (uint32_t)(p.ktot * ESZ) + (uint32_t)((((lane & 7) ^ (brow1 & 7))) * 16);
  const uint32_t lds_b = lds_base + (uint32_t)(NSUB * patch_stride) + (uint32_t)wave_u * 1024u;
#define HALO_ISSUE_B(BUF, KSTEP)                                                                             \
  {                                                                                                          \
    const uint32_t lb_ = lds_b + (uint32_t)(BUF) * (64 * 128);                                               \
    const int so_ = (KSTEP)*128;                                                                             \
    asm volatile(                                                                                            \
        "s_nop 4\n\t"                                                                                        \
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %4 offen lds\n\t"                         \
        "s_add_u32 m0, %2, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds" ::"v"(b_off0),      \
        "v"(b_off1), "s"(lb_), "s"(rs_b), "s"(so_)                                                           \
        : "memory", "scc");                                                                                  \
  }
  HALO_ISSUE_B(0, 0)

  f32x4 acc[4][NF];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave;  // sub-tile of this wave
  const int i16 = lane & 15, g = lane >> 4;
  const int py0 = i16 >> 3, px = i16 & 7;   // fragment f covers rows py = 2f + py0
  const int swb = i16 & 7;
  const unsigned char* a_patch = sA + wr * patch_stride;

  int kr = 0, ks = 0, cc = 0;
  for (int k = 0; k < nk; ++k) {
    const int buf = k & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (k + 1 < nk) HALO_ISSUE_B(buf ^ 1, k + 1)
    const int ky = p.dgrad ? (p.R - 1 - kr) : kr, kx = p.dgrad ? (p.S - 1 - ks) : ks;
    const int hx = px + kx;
    const unsigned char* a = a_patch + ((py0 + ky) * p.PW + hx) * PB + cc * 128;
    const unsigned char* b = sB + buf * (64 * 128) + i16 * 128;
    const int rowstep = 2 * p.PW * PB;  // fragment f+1 is two patch rows further down
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch = g + 4 * h;
      const int coffa = (swz ? (ch ^ (hx & 7)) : ch) << 4;
      const int coffb = (ch ^ swb) << 4;
      uint4 af[4], bfr[NF];
#pragma unroll
      for (int f = 0; f < 4; ++f) af[f] = *reinterpret_cast<const uint4*>(a + f * rowstep + coffa);
#pragma unroll
      for (int j = 0; j < NF; ++j) bfr[j] = *reinterpret_cast<const uint4*>(b + j * 16 * 128 + coffb);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[f]), __builtin_bit_cast(bf16x8, bfr[j]),
                                                                acc[f][j], 0, 0, 0);
          } else {
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].x), __uint_as_float(bfr[j].x), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].y), __uint_as_float(bfr[j].y), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].z), __uint_as_float(bfr[j].z), acc[f][j], 0, 0, 0);
            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].w), __uint_as_float(bfr[j].w), acc[f][j], 0, 0, 0);
          }
        }
    }
    if (++cc == cpt) {
      cc = 0;
      if (++ks == p.S) {
        ks = 0;
        ++kr;
      }
    }
  }
#undef HALO_ISSUE_B
  __syncthreads();

  // ---- epilogue (as igemm.hip, with the sub-tile row -> pixel mapping) ----
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) sC[(wr * 64 + f * 16 + g * 4 + reg) * LDC + j * 16 + i16] = acc[f][j][reg];
  __syncthreads();

  T* __restrict__ out = (T*)p.out;
  const T* __restrict__ resid = (const T*)p.resid;
  const T* __restrict__ mask = (const T*)p.mask;
  constexpr int TPR = BN / 8, RPP = 256 / TPR;
  const int col8 = (tid % TPR) * 8;
  const int n = col8;
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  if (n < p.co) {
    const bool vec = p.vec_ok && (n + 8 <= p.co);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (p.bias && n + e < p.co) ? p.bias[n + e] : 0.f;
    for (int r0 = tid / TPR; r0 < NSUB * 64; r0 += RPP) {
      const int st = sub0 + (r0 >> 6);
      if (st >= p.n_sub) break;
      const int img = st / p.tiles_per_img;
      const int trem = st - img * p.tiles_per_img;
      const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
      const size_t m = ((size_t)img * p.ho + ty * 8 + ((r0 >> 3) & 7)) * p.wo + tx * 8 + (r0 & 7);
      const size_t o = m * p.ldo + n;
      float v[8];
      const float4 c0v = *reinterpret_cast<const float4*>(sC + r0 * LDC + col8);
      const float4 c1v = *reinterpret_cast<const float4*>(sC + r0 * LDC + col8 + 4);
      v[0] = c0v.x + bv[0]; v[1] = c0v.y + bv[1]; v[2] = c0v.z + bv[2]; v[3] = c0v.w + bv[3];
      v[4] = c1v.x + bv[4]; v[5] = c1v.y + bv[5]; v[6] = c1v.z + bv[6]; v[7] = c1v.w + bv[7];
      if (vec) {
        T rv[8], mv[8], ov[8];
        if (resid) {
          if constexpr (ESZ == 2) *reinterpret_cast<uint4*>(rv) = *reinterpret_cast<const uint4*>(resid + o);
          else { reinterpret_cast<uint4*>(rv)[0] = reinterpret_cast<const uint4*>(resid + o)[0]; reinterpret_cast<uint4*>(rv)[1] = reinterpret_cast<const uint4*>(resid + o)[1]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += to_f32<T>(rv[e]);
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mask) {
          if constexpr (ESZ == 2) *reinterpret_cast<uint4*>(mv) = *reinterpret_cast<const uint4*>(mask + o);
          else { reinterpret_cast<uint4*>(mv)[0] = reinterpret_cast<const uint4*>(mask + o)[0]; reinterpret_cast<uint4*>(mv)[1] = reinterpret_cast<const uint4*>(mask + o)[1]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (to_f32<T>(mv[e]) > 0.f) ? v[e] : 0.f;
        }
        if (out) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ov[e] = from_f32<T>(v[e]);
            cs[e] += to_f32<T>(ov[e]);
          }
          if constexpr (ESZ == 2) *reinterpret_cast<uint4*>(out + o) = *reinterpret_cast<const uint4*>(ov);
          else { reinterpret_cast<uint4*>(out + o)[0] = reinterpret_cast<const uint4*>(ov)[0]; reinterpret_cast<uint4*>(out + o)[1] = reinterpret_cast<const uint4*>(ov)[1]; }
        }
        if (p.out_f32) {
          *reinterpret_cast<float4*>(p.out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(p.out_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      } else {
        for (int e = 0; e < 8 && n + e < p.co; ++e) {
          float x = v[e];
          if (resid) x += to_f32<T>(resid[o + e]);
          if (p.relu) x = fmaxf(x, 0.f);
          if (mask) x = (to_f32<T>(mask[o + e]) > 0.f) ? x : 0.f;
          if (out) {
            out[o + e] = from_f32<T>(x);
            cs[e] += to_f32<T>(from_f32<T>(x));
          }
          if (p.out_f32) p.out_f32[o + e] = x;
        }
      }
    }
  }
  if (p.colsum_part) {  // per-block column sums; the consumer only needs their total, not a row-tile geometry
    float* sR = reinterpret_cast<float*>(smem + NSUB * 64 * LDC * 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) sR[(tid / TPR) * BN + col8 + e] = cs[e];
    __syncthreads();
    if (tid < BN && tid < p.co) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) t += sR[r * BN + tid];
      // the consumer sums ceil(rows/128) entries: this block covers two of them
      p.colsum_part[(size_t)(2 * blockIdx.x) * p.ldo + tid] = t;
      if ((2 * blockIdx.x + 1) * 2 < (unsigned)p.n_sub) p.colsum_part[(size_t)(2 * blockIdx.x + 1) * p.ldo + tid] = 0.f;
    }
  }
}

}  // namespace

// Returns 1 if the halo kernel took the launch, 0 if the shape is not eligible, < 0 on error.
int vdqn_try_halo_conv(const vdqn_conv_args* a, hipStream_t st) {
  static const bool disabled = [] {
    const char* e = getenv("VDQN_NO_HALO");
    return e && e[0] == '1';
  }();
  if (disabled) return 0;
  const int esz = a->dtype == VDQN_BF16 ? 2 : 4;
  if (a->stride != 1 || a->co != 64 || a->ci != 64 || (a->ho % 8) || (a->wo % 8)) return 0;
  const bool layer = a->r == 3 && a->s == 3 && a->pad == 1 && a->pix_stride == 64 && a->hi == a->ho && a->wi == a->wo;
  const bool stem = a->r == 4 && a->s == 1 && a->pad == 0 && a->pix_stride == 16 && a->mode == 0 && a->hi == a->ho + 3 && a->wi == a->wo + 3;
  if (!layer && !stem) return 0;
  HaloParams p;
  memset(&p, 0, sizeof(p));
  p.in = a->in; p.wt = a->wt; p.bias = a->bias; p.resid = a->resid; p.mask = a->mask; p.out = a->out; p.out_f32 = a->out_f32;
  p.colsum_part = a->colsum_part;
  p.n_img = a->n_img; p.hi = a->hi; p.wi = a->wi; p.pix_stride = a->pix_stride; p.ho = a->ho; p.wo = a->wo; p.co = a->co; p.ldo = a->ldo;
  p.R = a->r; p.S = a->s; p.pad = a->pad; p.relu = a->relu; p.dgrad = a->mode == 1;
  p.span_px = a->ci / a->pix_stride;
  p.PW = 8 + (a->s - 1) + (p.span_px - 1);
  p.PH = 8 + a->r - 1;
  p.tiles_x = a->wo / 8;
  p.tiles_per_img = (a->ho / 8) * p.tiles_x;
  p.n_sub = a->n_img * p.tiles_per_img;
  p.ktot = a->r * a->s * a->ci;
  p.in_bytes = (long long)a->n_img * a->hi * a->wi * a->pix_stride * esz;
  p.wt_bytes = 64 * p.ktot * esz;
  const uintptr_t al = (uintptr_t)a->out | (uintptr_t)a->out_f32 | (uintptr_t)a->resid | (uintptr_t)a->mask;
  p.vec_ok = ((a->ldo * esz) % 16 == 0) && ((al & 15) == 0) && (a->ldo % 8 == 0);
  if (2ll * a->hi * a->wi * a->pix_stride * esz >= 0x7fffffffLL) return 0;
  const int patch_stride = ((p.PH * p.PW * a->pix_stride * esz + 1023) / 1024) * 1024;
  const size_t main_bytes = 4 * (size_t)patch_stride + 2 * 64 * 128, epi_bytes = 256 * 68 * 4 + 256 * 8 * 4;
  const size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  if (smem > 160 * 1024) return 0;
  const unsigned grid = (unsigned)((p.n_sub + 3) / 4);
  const double M = (double)a->n_img * a->ho * a->wo;
  vdqn_prof_begin(a->dtype == VDQN_BF16 ? "igemm_halo<bf16>" : "igemm_halo<f32>", 2.0 * M * a->co * p.ktot,
                  esz * ((double)a->n_img * a->hi * a->wi * a->pix_stride + 64.0 * p.ktot + M * a->co * (1 + (a->resid != nullptr) + (a->mask != nullptr))), st);
  if (a->dtype == VDQN_BF16) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_kernel<bf16raw>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    hipLaunchKernelGGL((igemm_halo_kernel<bf16raw>), dim3(grid), dim3(256), smem, st, p);
  } else {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    hipLaunchKernelGGL((igemm_halo_kernel<float>), dim3(grid), dim3(256), smem, st, p);
  }
  vdqn_prof_end(st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vdqn_set_error("igemm_halo launch failed: %s", hipGetErrorString(e));
    return VDQN_ERR_LAUNCH;
  }
  return 1;
}
