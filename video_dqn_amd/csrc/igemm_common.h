// Shared pieces of the implicit-GEMM kernels (igemm.hip, win9.hip): launch parameters, the LDS-free epilogue, the
// MFMA / VALU / LDS interleave pattern.  Included inside each file's anonymous namespace user — everything here is
// `static`/template/inline device code.
#pragma once
#include "common.h"

namespace {


typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
  const void* in;
  const void* wt;
  const float* bias;
  const void* resid;
  const void* mask;
  void* out;
  float* out_f32;
  float* colsum_part;
  void* pool_out;      // MODE 3 (stem): max-pooled output [n][56][56][64] and its argmax codes
  uint8_t* pool_idx;
  int n_idx_img;  // MODE 3: arg-max bytes for images [0, n_idx_img) only
  int n_img, hi, wi, ci, pix_stride, ho, wo, co, ldo, r, s, stride, pad, relu;
  int M, howo, ktot, nk, tiles_m, tiles_n;
  int cls_tile0[5], cls_h[2], cls_w[2];  // MODE 2: first tile of each output-parity class; class heights / widths
  int cls_interleave;                    // MODE 2: the four classes have equal tile counts and tile t belongs to class t & 3
  long long in_bytes;
  int wt_bytes;
  int vec_ok;
  // sibling 1x1 convolution fused into a 3x3 / stride-2 / pad-1 launch (ResNet downsample branch; generic kernel only).
  // Forward (MODE 0): the column tiles tiles_n1 .. tiles_n - 1 compute out2 = conv1x1_stride2(in) with wt2 / bias2 / relu2 —
  // the 1x1's input pixel is the 3x3's centre tap, so both read the same staged rows.  Stride-2 data gradient (MODE 2): the
  // tiles of output-parity class (0, 0) run ci2 / 64 extra K-steps over in2 (the gradient of the 1x1's output, same geometry as
  // `in`) and wt2, adding the 1x1's data gradient in the accumulators (no intermediate tensor).
  const void* in2;
  const void* wt2;
  const float* bias2;
  void* out2;
  int co2, ldo2, relu2, ci2, wt2_bytes, tiles_n1;
  int no_lean;  // VDQN_LEAN_EPILOGUE=0 (A/B switch): the window kernels keep igemm_epilogue where the lean one would serve
};

constexpr unsigned kOob = 0x80000000u;  // voffset beyond any descriptor range (num_records <= 0x7fffffff)

// ---- epilogue, straight from the accumulators (shared by the generic and the window kernel): lane (i16, g) owns channels
// [ncol, ncol + CPL) of pixels f*16 + i16 of its wave's 64 rows ----
template <typename T, int BM, int BN, int MODE, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[4][BN / (16 * WN)], unsigned char* smem, int m0, int n0, int tile_m,
                                               int rows_total, int pix_per_img, int row_w, int cls_ph, int cls_pw, const float* __restrict__ bias,
                                               int tid_override = -1) {
  constexpr int ESZ = (int)sizeof(T);
  constexpr int NF = BN / (16 * WN);
  constexpr int CPL = 4 * NF;
  // tid_override: kernels whose workgroup holds two independent 4-wave groups pass the thread's index inside its group
  // (smem is then that group's LDS region; the barriers below are workgroup-wide, which both groups reach equally often)
  const int tid = tid_override >= 0 ? tid_override : (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  T* __restrict__ out = (T*)p.out;
  const T* __restrict__ resid = (const T*)p.resid;
  const T* __restrict__ mask = (const T*)p.mask;
  constexpr int V16 = CPL * ESZ / 16;  // 16-byte vectors per lane and pixel
  const int ncol = n0 + wc * (BN / WN) + g * CPL;
  float cs[CPL];  // per-lane column sums of the values this tile stores (for the BN-shift / bias gradient)
#pragma unroll
  for (int e = 0; e < CPL; ++e) cs[e] = 0.f;
  if (ncol < p.co) {
    const bool vec = p.vec_ok && (ncol + CPL <= p.co);
    float bv[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) bv[e] = (bias && ncol + e < p.co) ? bias[ncol + e] : 0.f;
    size_t o[4];
    bool okr[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      int m = m0 + wr * 64 + f * 16 + i16;
      okr[f] = m < rows_total;
      if constexpr (MODE == 2) {  // class-local row -> output pixel
        const int mm = okr[f] ? m : m0;
        const int img = mm / pix_per_img;
        const int rem = mm - img * pix_per_img;
        const int ohc = rem / row_w;
        m = (img * p.ho + 2 * ohc + cls_ph) * p.wo + 2 * (rem - ohc * row_w) + cls_pw;
      }
      o[f] = (size_t)m * p.ldo + ncol;
    }
    if (vec) {
      uint4 rv[4][V16], mv[4][V16];
      if (resid) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int q = 0; q < V16; ++q) rv[f][q] = okr[f] ? reinterpret_cast<const uint4*>(resid + o[f])[q] : make_uint4(0, 0, 0, 0);
      }
      if (mask) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int q = 0; q < V16; ++q) mv[f][q] = okr[f] ? reinterpret_cast<const uint4*>(mask + o[f])[q] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        float v[CPL];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[f][j][r] + bv[j * 4 + r];
        if (resid) {
          const T* pr = reinterpret_cast<const T*>(rv[f]);
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] += to_f32<T>(pr[e]);
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mask) {
          const T* pm = reinterpret_cast<const T*>(mv[f]);
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] = (to_f32<T>(pm[e]) > 0.f) ? v[e] : 0.f;
        }
        if (okr[f]) {
          if (out) {
            T ov[CPL];
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
              ov[e] = from_f32<T>(v[e]);
              cs[e] += to_f32<T>(ov[e]);
            }
#pragma unroll
            for (int q = 0; q < V16; ++q) reinterpret_cast<uint4*>(out + o[f])[q] = reinterpret_cast<const uint4*>(ov)[q];
          }
          if (p.out_f32) {
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q)
              *reinterpret_cast<float4*>(p.out_f32 + o[f] + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          }
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (!okr[f]) continue;
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = j * 4 + r;
            if (ncol + e >= p.co) continue;
            float x = acc[f][j][r] + bv[e];
            if (resid) x += to_f32<T>(resid[o[f] + e]);
            if (p.relu) x = fmaxf(x, 0.f);
            if (mask) x = (to_f32<T>(mask[o[f] + e]) > 0.f) ? x : 0.f;
            if (out) {
              out[o[f] + e] = from_f32<T>(x);
              cs[e] += to_f32<T>(from_f32<T>(x));
            }
            if (p.out_f32) p.out_f32[o[f] + e] = x;
          }
      }
    }
  }
  if (p.colsum_part) {  // uniform branch: partial column sums of this tile -> colsum_part[tile_m][ldo]
    // sum the 16 pixel-lanes of every channel group, then the BM/64 wave rows through LDS
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
      float t = cs[e];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      t += __shfl_xor(t, 8, 64);
      cs[e] = t;
    }
    __syncthreads();  // every wave is past its last fragment read: LDS can be reused
    float* sR = reinterpret_cast<float*>(smem);
    if (i16 == 0) {
#pragma unroll
      for (int e = 0; e < CPL; ++e) sR[wr * BN + wc * (BN / WN) + g * CPL + e] = cs[e];
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


// ---- lean forward epilogue of the 128 x 128 window kernels (round 5): bias + optional residual + optional ReLU -> bf16, whole
// 16-byte vectors.  igemm_epilogue above serves every kernel and operand combination; in the persistent window kernels it cost
// ~8 k cycles per tile (tools/stamp_s2.py, profiles/r05b_stamp_s2.txt): sixteen single-dword bias loads, 64-bit address arithmetic,
// per-element branches of the ragged path, and hundreds of v_readlane reloads of spilled scalars.  Here: buffer loads / stores with
// 32-bit offsets (a row behind rows_end gets an out-of-range offset: its loads return 0, its stores are dropped), the bias as four
// 16-byte loads, every load issued before the first is used.  Same arithmetic in the same order as igemm_epilogue's vector path:
// (acc + bias) + resid, max(., 0), round to bf16 — the same bits.
// The caller guarantees: bf16, CPL = 16, ncol + 16 <= co, ldo % 8 == 0, M * ldo * 2 < 2^31, bias != nullptr and 16-byte aligned. ----
struct LeanEpi {
  __amdgpu_buffer_rsrc_t out, res, bias;
  int ldo, relu, has_res;
};
__device__ __forceinline__ LeanEpi make_lean_epi(void* out, const void* resid, const float* bias, long long rows, int ldo, int co, int relu) {
  LeanEpi e;
  const int bytes = (int)(rows * ldo * 2);
  e.out = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
  e.res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(resid), 0, resid ? bytes : 0, 0x00020000);
  e.bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, co * 4, 0x00020000);
  e.ldo = ldo; e.relu = relu; e.has_res = resid != nullptr;
  return e;
}
// The loads of a tile's epilogue (bias, residual) and the offsets, split from the arithmetic so that a kernel can issue them before
// its last K-step's MFMAs have drained (win9u_kernel: in place of the fragment reads nobody uses in a tile's last step).
struct LeanPre {
  u32x4 bq[4], rv[4][2];
  uint32_t off[4];
};
template <int WN>
__device__ __forceinline__ void lean_prefetch_128(const LeanEpi& e, LeanPre& pre, int m0, int n0, int rows_end, int tid) {
  static_assert(WN == 2, "64 x 64 wave tiles: 16 consecutive channels per lane");
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  const int ncol = n0 + wc * 64 + g * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) pre.bq[q] = __builtin_amdgcn_raw_buffer_load_b128(e.bias, ncol * 4 + 16 * q, 0, 0);
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int m = m0 + wr * 64 + f * 16 + i16;
    pre.off[f] = m < rows_end ? (uint32_t)(m * e.ldo + ncol) * 2u : kOob;
  }
  if (e.has_res) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      pre.rv[f][0] = __builtin_amdgcn_raw_buffer_load_b128(e.res, (int)pre.off[f], 0, 0);
      pre.rv[f][1] = __builtin_amdgcn_raw_buffer_load_b128(e.res, (int)pre.off[f] + 16, 0, 0);
    }
  }
}
__device__ __forceinline__ void lean_finish_128(const LeanEpi& e, f32x4 (&acc)[4][4], const LeanPre& pre) {
  const float* bv = reinterpret_cast<const float*>(pre.bq);
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[f][j][r] + bv[j * 4 + r];
    if (e.has_res) {
      const bf16raw* pr = reinterpret_cast<const bf16raw*>(pre.rv[f]);
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] += bf16_to_f32(pr[k]);
    }
    if (e.relu) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    bf16raw ov[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ov[k] = f32_to_bf16(v[k]);
    __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const u32x4*>(ov)[0], e.out, (int)pre.off[f], 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const u32x4*>(ov)[1], e.out, (int)pre.off[f] + 16, 0, 0);
  }
}
template <int WN>
__device__ __forceinline__ void lean_epilogue_128(const LeanEpi& e, f32x4 (&acc)[4][128 / (16 * WN)], int m0, int n0, int rows_end, int tid) {
  LeanPre pre;
  lean_prefetch_128<WN>(e, pre, m0, n0, rows_end, tid);
  lean_finish_128(e, acc, pre);
}

// The data-gradient form: gx = mask > 0 ? (acc + resid) : 0 -> bf16, and the per-tile column sums of what was stored (the bias /
// BatchNorm-shift gradient of the layer below): igemm_epilogue's arithmetic (its `+ 0.f` for the absent bias included: -0 -> +0)
// with buffer loads / stores.  scratch: 2 x 128 floats of LDS no wave still reads.
struct LeanEpiD {
  __amdgpu_buffer_rsrc_t out, res, msk;
  float* colsum_part;
  int ldo, has_res, has_msk, co;
  int rows;  // rows of `out` (the last 256-row tile of a launch may hold one 128-row half only: one column-sum entry)
};
__device__ __forceinline__ LeanEpiD make_lean_epi_d(void* out, const void* resid, const void* mask, float* colsum_part, long long rows, int ldo, int co) {
  LeanEpiD e;
  const int bytes = (int)(rows * ldo * 2);
  e.out = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
  e.res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(resid), 0, resid ? bytes : 0, 0x00020000);
  e.msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mask), 0, mask ? bytes : 0, 0x00020000);
  e.colsum_part = colsum_part;
  e.ldo = ldo; e.has_res = resid != nullptr; e.has_msk = mask != nullptr; e.co = co;
  e.rows = (int)rows;
  return e;
}
// NF = 16-wide channel fragments per wave (4: 128-column tiles, 2: 64-column tiles; WN = 2 waves along N either way).  off[f]: byte
// offset of the lane's pixel f in out / resid / mask at channel ncol, kOob for rows that do not exist.
// NWR = wave rows of the workgroup (2: 128-row tiles; 4: 256-row tiles, whose two 128-row halves write entries 2 tile_m and 2 tile_m + 1
// of the column sums; scratch: NWR x BN floats)
template <int NF, int NWR = 2>
__device__ __forceinline__ void lean_epilogue_dgrad(const LeanEpiD& e, f32x4 (&acc)[4][NF], float* scratch, const uint32_t (&off)[4], int n0, int tile_m, int tid) {
  constexpr int CPL = 4 * NF, BN = 32 * NF, V = CPL / 8;  // channels per lane, tile columns, 16-byte vectors per pixel
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, g = lane >> 4;
  u32x4 rv[4][V], mv[4][V];
  if (e.has_res) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int q = 0; q < V; ++q) rv[f][q] = __builtin_amdgcn_raw_buffer_load_b128(e.res, (int)off[f] + 16 * q, 0, 0);
  }
  if (e.has_msk) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int q = 0; q < V; ++q) mv[f][q] = __builtin_amdgcn_raw_buffer_load_b128(e.msk, (int)off[f] + 16 * q, 0, 0);
  }
  float cs[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) cs[k] = 0.f;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    float v[CPL];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[f][j][r] + 0.f;
    if (e.has_res) {
      const bf16raw* pr = reinterpret_cast<const bf16raw*>(rv[f]);
#pragma unroll
      for (int k = 0; k < CPL; ++k) v[k] += bf16_to_f32(pr[k]);
    }
    if (e.has_msk) {
      const bf16raw* pm = reinterpret_cast<const bf16raw*>(mv[f]);
#pragma unroll
      for (int k = 0; k < CPL; ++k) v[k] = (bf16_to_f32(pm[k]) > 0.f) ? v[k] : 0.f;
    }
    bf16raw ov[CPL];
    const float live = off[f] != kOob ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      ov[k] = f32_to_bf16(v[k]);
      cs[k] += live * bf16_to_f32(ov[k]);
    }
#pragma unroll
    for (int q = 0; q < V; ++q) __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const u32x4*>(ov)[q], e.out, (int)off[f] + 16 * q, 0, 0);
  }
  if (e.colsum_part) {  // uniform: the 16 pixel-lanes of every channel, then the two wave rows through LDS (igemm_epilogue's order)
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      float t = cs[k];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      t += __shfl_xor(t, 8, 64);
      cs[k] = t;
    }
    __syncthreads();
    if (i16 == 0) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) scratch[wr * BN + wc * (BN / 2) + g * CPL + k] = cs[k];
    }
    __syncthreads();
    if constexpr (NWR == 2) {
      if (tid < BN && n0 + tid < e.co) e.colsum_part[(size_t)tile_m * e.ldo + n0 + tid] = scratch[tid] + scratch[BN + tid];
    } else {
      const int half = tid / BN, col = tid - half * BN;
      if (tid < (NWR / 2) * BN && n0 + col < e.co && (tile_m * (NWR / 2) + half) * 128 < e.rows)
        e.colsum_part[(size_t)(tile_m * (NWR / 2) + half) * e.ldo + n0 + col] = scratch[2 * half * BN + col] + scratch[(2 * half + 1) * BN + col];
    }
  }
}
template <int WN, int NWR = 2>
__device__ __forceinline__ void lean_epilogue_dgrad_128(const LeanEpiD& e, f32x4 (&acc)[4][128 / (16 * WN)], float* scratch, int m0, int n0, int tile_m,
                                                        int rows_end, int tid) {
  static_assert(WN == 2, "64 x 64 wave tiles: 16 consecutive channels per lane");
  const int lane = tid & 63, wave = tid >> 6;
  const int ncol = n0 + (wave & 1) * 64 + (lane >> 4) * 16;
  uint32_t off[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int m = m0 + (wave >> 1) * 64 + f * 16 + (lane & 15);
    off[f] = m < rows_end ? (uint32_t)(m * e.ldo + ncol) * 2u : kOob;
  }
  lean_epilogue_dgrad<4, NWR>(e, acc, scratch, off, n0, tile_m, tid);
}

#define VDQN_INTERLEAVE(N)                                  \
  if constexpr (sizeof(T) == 2) {                           \
    _Pragma("unroll") for (int g_ = 0; g_ < (N); ++g_) {    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    \
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);    \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
    }                                                       \
  }

}  // namespace
