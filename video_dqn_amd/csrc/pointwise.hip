// HBM-bound kernels of the hot path: input packing (normalise + space-to-depth), 3x3/2 max-pool forward and
// backward, the fused Double-DQN target / TD-loss / dQ kernel, and the flat fused Adam.
// All accesses are 16-byte vectors over the contiguous NHWC channel axis.
#include <math.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// pack_input: 224x224x3 frame -> [115][115][16] space-to-depth operand of the 4x4/1 stem GEMM.
// dst pixel (y, x) holds source pixels (2(y-2)+bh, 2(x-2)+bw), channel (bh*2+bw)*3+c; channels 12..15 = 0.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const void* __restrict__ src, int src_kind, T* __restrict__ dst, int n_img) {
  const int total = n_img * 115 * 115;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / (115 * 115);
    const int rem = i - n * (115 * 115);
    const int y = rem / 115, x = rem - y * 115;
    const int ys = y - 2, xs = x - 2;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
    if ((unsigned)ys < 112u && (unsigned)xs < 112u) {
#pragma unroll
      for (int bh = 0; bh < 2; ++bh)
#pragma unroll
        for (int bw = 0; bw < 2; ++bw) {
          const int Y = 2 * ys + bh, X = 2 * xs + bw;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float val;
            if (src_kind == 0) {
              const uint8_t u = ((const uint8_t*)src)[(((size_t)n * 224 + Y) * 224 + X) * 3 + c];
              val = (((float)u / 255.0f) - mean[c]) / stdv[c];
            } else {
              val = ((const float*)src)[(((size_t)n * 3 + c) * 224 + Y) * 224 + X];
            }
            v[(bh * 2 + bw) * 3 + c] = val;
          }
        }
    }
    T* d = dst + (size_t)i * 16;
    if constexpr (sizeof(T) == 2) {
      uint32_t w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
      reinterpret_cast<uint4*>(d)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(d)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) reinterpret_cast<float4*>(d)[e] = make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
    }
  }
}

// uint8 HWC source: a block walks kPackPairs pairs of packed rows; per pair the four source rows (672 bytes each) are read as 16-byte
// vectors into LDS and every thread builds one packed pixel from there.  The normalisation (u / 255 - mean) / std — two IEEE
// divisions per element, 24 per packed pixel: what bounded this kernel at 2.9 TB/s — is evaluated ONCE per block for the 3 x 256
// possible (channel, byte) pairs into an LDS table, with the same expression: same bits as the per-element arithmetic.
constexpr int kPackPairs = 4;
template <typename T>
__global__ __launch_bounds__(256) void pack_input_rows_kernel(const uint8_t* __restrict__ src, T* __restrict__ dst) {
  __shared__ uint4 rows[4][42];
  __shared__ T lut[3][256];
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  {
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) lut[c][tid] = from_f32<T>((((float)tid / 255.0f) - mean[c]) / stdv[c]);
  }
  const int yy = tid / 115, x = tid - yy * 115;
  for (int pr = 0; pr < kPackPairs; ++pr) {
    const int y0 = 2 * ((int)blockIdx.x * kPackPairs + pr);  // packed rows y0, y0 + 1 -> source rows 2 (y0 - 2) .. + 3
    if (y0 >= 115) break;
    const int Y0 = 2 * (y0 - 2);
    __syncthreads();  // the previous pair's readers are done (first pass: the table is written)
    if (tid < 168) {
      const int r = tid / 42, c = tid - r * 42, Y = Y0 + r;
      rows[r][c] = ((unsigned)Y < 224u) ? reinterpret_cast<const uint4*>(src + ((size_t)n * 224 + Y) * 672)[c] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int y = y0 + yy;
    if (tid >= 230 || y >= 115) continue;
    const int ys = y - 2, xs = x - 2;
    __attribute__((aligned(16))) T v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = from_f32<T>(0.f);
    if ((unsigned)ys < 112u && (unsigned)xs < 112u) {
      const uint8_t* b0 = reinterpret_cast<const uint8_t*>(rows[2 * yy]) + 6 * xs;
      const uint8_t* b1 = reinterpret_cast<const uint8_t*>(rows[2 * yy + 1]) + 6 * xs;
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        v[e] = lut[e % 3][b0[e]];
        v[6 + e] = lut[e % 3][b1[e]];
      }
    }
    T* d = dst + ((size_t)n * 115 * 115 + (size_t)y * 115 + x) * 16;
    constexpr int V16 = (int)(16 * sizeof(T) / 16);
#pragma unroll
    for (int q = 0; q < V16; ++q) reinterpret_cast<uint4*>(d)[q] = reinterpret_cast<const uint4*>(v)[q];
  }
}

// ---------------------------------------------------------------------------------------------------------
// max-pool 3x3 / stride 2 / pad 1 (NHWC), first maximum wins (torch CPU semantics)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, uint8_t* __restrict__ idx,
                                                          int n_img, int hi, int wi, int c) {
  constexpr int E16 = 16 / (int)sizeof(T);
  const int ho = (hi + 2 - 3) / 2 + 1, wo = (wi + 2 - 3) / 2 + 1;
  const int cg_n = c / E16;
  const long total = (long)n_img * ho * wo * cg_n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_n);
    long t = i / cg_n;
    const int ow = (int)(t % wo);
    t /= wo;
    const int oh = (int)(t % ho);
    const int n = (int)(t / ho);
    float best[E16];
    uint8_t bi[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = oh * 2 - 1 + kh;
      if ((unsigned)h >= (unsigned)hi) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = ow * 2 - 1 + kw;
        if ((unsigned)w >= (unsigned)wi) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(in + (((size_t)n * hi + h) * wi + w) * c + cg * E16);
        const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          const float f = to_f32<T>(pv[e]);
          if (first || f > best[e] || f != f) {
            best[e] = f;
            bi[e] = (uint8_t)(kh * 3 + kw);
          }
        }
        first = false;
      }
    }
    const size_t o = (((size_t)n * ho + oh) * wo + ow) * c + cg * E16;
    T ov[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) ov[e] = from_f32<T>(best[e]);
    *reinterpret_cast<uint4*>(out + o) = *reinterpret_cast<const uint4*>(ov);
    if constexpr (E16 == 8) *reinterpret_cast<uint2*>(idx + o) = *reinterpret_cast<const uint2*>(bi);
    else *reinterpret_cast<uint32_t*>(idx + o) = *reinterpret_cast<const uint32_t*>(bi);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ gy, const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                                          T* __restrict__ gx, int n_img, int hi, int wi, int c) {
  constexpr int E16 = 16 / (int)sizeof(T);
  const int ho = (hi + 2 - 3) / 2 + 1, wo = (wi + 2 - 3) / 2 + 1;
  const int cg_n = c / E16;
  const long total = (long)n_img * hi * wi * cg_n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_n);
    long t = i / cg_n;
    const int w = (int)(t % wi);
    t /= wi;
    const int h = (int)(t % hi);
    const int n = (int)(t / hi);
    float s[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) s[e] = 0.f;
    const int oh_lo = h / 2, oh_hi = min(ho - 1, (h + 1) / 2);
    const int ow_lo = w / 2, ow_hi = min(wo - 1, (w + 1) / 2);
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int tap = (h - (2 * oh - 1)) * 3 + (w - (2 * ow - 1));
        const size_t o = (((size_t)n * ho + oh) * wo + ow) * c + cg * E16;
        const uint4 gv = *reinterpret_cast<const uint4*>(gy + o);
        const T* pg = reinterpret_cast<const T*>(&gv);
        uint8_t iv[E16];
        if constexpr (E16 == 8) *reinterpret_cast<uint2*>(iv) = *reinterpret_cast<const uint2*>(idx + o);
        else *reinterpret_cast<uint32_t*>(iv) = *reinterpret_cast<const uint32_t*>(idx + o);
#pragma unroll
        for (int e = 0; e < E16; ++e)
          if (iv[e] == tap) s[e] += to_f32<T>(pg[e]);
      }
    const size_t xo = (((size_t)n * hi + h) * wi + w) * c + cg * E16;
    T ov[E16];
    if (x) {
      const uint4 xv = *reinterpret_cast<const uint4*>(x + xo);
      const T* px = reinterpret_cast<const T*>(&xv);
#pragma unroll
      for (int e = 0; e < E16; ++e) ov[e] = from_f32<T>(to_f32<T>(px[e]) > 0.f ? s[e] : 0.f);
    } else {
#pragma unroll
      for (int e = 0; e < E16; ++e) ov[e] = from_f32<T>(s[e]);
    }
    *reinterpret_cast<uint4*>(gx + xo) = *reinterpret_cast<const uint4*>(ov);
  }
}

// Row-pair tiled variant (even input heights, c * sizeof(T) a multiple of 16): one block produces input rows 2k and 2k+1
// of one image.  Row 2k only receives from pooled row k (kh = 1), row 2k+1 from pooled rows k (kh = 2) and k+1 (kh = 0),
// so the block stages those two pooled rows of gy and idx in LDS once (each pooled element is then fetched from HBM/L2
// by two blocks instead of by 2.25 scattered gathers) and writes full 16-byte vectors of gx.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_rows_kernel(const T* __restrict__ gy, const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                                               T* __restrict__ gx, int hi, int wi, int c) {
  constexpr int E16 = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char mp_smem[];
  const int ho = hi / 2, wo = wi / 2;
  const int cg_n = c / E16;
  const int n = blockIdx.y, k = blockIdx.x;
  const int rows = (k + 1 < ho) ? 2 : 1;          // pooled rows k (and k + 1)
  const int vec_row = wo * cg_n;                    // 16-byte gy vectors per pooled row
  uint4* sG = reinterpret_cast<uint4*>(mp_smem);    // [2][wo][cg_n] gy vectors
  uint8_t* sI = mp_smem + 2 * vec_row * 16;         // [2][wo][c] arg-max codes
  const size_t prow = ((size_t)n * ho + k) * wo;    // first pooled pixel of row k
  for (int i = threadIdx.x; i < rows * vec_row; i += 256) {
    sG[i] = reinterpret_cast<const uint4*>(gy + prow * c)[i];
    if constexpr (E16 == 8) reinterpret_cast<uint2*>(sI)[i] = reinterpret_cast<const uint2*>(idx + prow * c)[i];
    else reinterpret_cast<uint32_t*>(sI)[i] = reinterpret_cast<const uint32_t*>(idx + prow * c)[i];
  }
  __syncthreads();
  const int items = 2 * wi * cg_n;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int cg = it % cg_n;
    const int t = it / cg_n;
    const int w = t % wi, hr = t / wi;              // hr = 0: row 2k, 1: row 2k + 1
    const int h = 2 * k + hr;
    float s[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) s[e] = 0.f;
    const int ow_lo = w / 2, ow_hi = min(wo - 1, (w + 1) / 2);
    const int pr_hi = hr == 0 ? 0 : rows - 1;       // pooled rows (relative to k) whose window holds h
    for (int pr = 0; pr <= pr_hi; ++pr) {
      const int kh = h - (2 * (k + pr) - 1);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int tap = kh * 3 + (w - (2 * ow - 1));
        const uint4 gv = sG[(pr * wo + ow) * cg_n + cg];
        const T* pg = reinterpret_cast<const T*>(&gv);
        const uint8_t* iv = sI + ((size_t)(pr * wo + ow) * cg_n + cg) * E16;
#pragma unroll
        for (int e = 0; e < E16; ++e)
          if (iv[e] == tap) s[e] += to_f32<T>(pg[e]);
      }
    }
    const size_t xo = (((size_t)n * hi + h) * wi + w) * c + cg * E16;
    T ov[E16];
    if (x) {
      const uint4 xv = *reinterpret_cast<const uint4*>(x + xo);
      const T* px = reinterpret_cast<const T*>(&xv);
#pragma unroll
      for (int e = 0; e < E16; ++e) ov[e] = from_f32<T>(to_f32<T>(px[e]) > 0.f ? s[e] : 0.f);
    } else {
#pragma unroll
      for (int e = 0; e < E16; ++e) ov[e] = from_f32<T>(s[e]);
    }
    *reinterpret_cast<uint4*>(gx + xo) = *reinterpret_cast<const uint4*>(ov);
  }
}

// ---------------------------------------------------------------------------------------------------------
// fused Double-DQN target + TD loss + dQ   (train_q_network.py:134-169,180)
// one thread per (sample, padded column); loss reduced per block, one atomic per block
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void td_loss_kernel(const vdqn_td_args a) {
  const int total = a.batch * a.ldq;
  float my_loss = 0.f;
  // one element per thread; a deterministic launch is ONE block that walks all elements (fixed summation order)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / a.ldq, col = i - b * a.ldq;
    float g = 0.f;
    if (col < a.n_cat * a.n_act) {
      const int c = col / a.n_act, ac = col - c * a.n_act;
      const int act = (int)a.act[b];
      if (a.q_copy) a.q_copy[(size_t)b * (a.n_cat * a.n_act) + col] = a.q_before[(size_t)b * a.ldq + col];
      if (ac == act) {
        const float qb = a.q_before[(size_t)b * a.ldq + col];
        const float* qo = a.q_after_online + (size_t)b * a.ldq + c * a.n_act;
        int best = 0;
        float bv = qo[0];
        for (int k = 1; k < a.n_act; ++k) {
          const float v = qo[k];
          if (v > bv) {  // strict: first maximum wins (torch.argmax)
            bv = v;
            best = k;
          }
        }
        float qa = a.q_after_target[(size_t)b * a.ldq + c * a.n_act + best];
        qa = qa * (1.0f - a.term[b * a.n_cat + c]);
        const float r = a.rew[b * a.n_cat + c];
        float y = a.linear ? r + (qa - 0.1f) : r + a.gamma * qa;
        if (a.clip_rect) y = fminf(fmaxf(y, 0.f), 1.f);
        const float d = qb - y;
        const float vm = a.use_valid ? a.valid[b * a.n_cat + c] : 1.0f;
        if (a.loss_kind == 1) {  // Huber, beta = 1 (torch.nn.functional.smooth_l1_loss)
          const float ad = fabsf(d);
          my_loss += (ad < 1.0f ? 0.5f * d * d : ad - 0.5f) * vm;
          g = fminf(fmaxf(d, -1.0f), 1.0f) * vm * a.inv_count;
        } else {
          my_loss += 0.5f * d * d * vm;
          g = d * vm * a.inv_count;
        }
      }
    }
    if (a.dq) ((T*)a.dq)[i] = from_f32<T>(g);
    if (a.dq_f32) a.dq_f32[i] = g;
  }
  // block reduction of the loss
  __shared__ float red[4];
  float v = my_loss;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = red[0] + red[1] + red[2] + red[3];
    if (s != 0.f) atomicAdd(a.loss, s * a.inv_count);
  }
}

// ground-truth branch (train_q_network.py:170-178)
template <typename T>
__global__ __launch_bounds__(256) void gt_loss_kernel(const float* __restrict__ q_before, const int64_t* __restrict__ act,
                                                      const float* __restrict__ gt, float* loss, T* dq, float* dq_f32, int batch,
                                                      int n_cat, int n_act, int ldq, float inv_count, int value_learning) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = batch * ldq;
  float my_loss = 0.f;
  if (i < total) {
    const int b = i / ldq, col = i - b * ldq;
    float g = 0.f;
    if (col < n_cat * n_act) {
      const int c = col / n_act, ac = col - c * n_act;
      if (ac == (int)act[b]) {
        const float qb = q_before[(size_t)b * ldq + col];
        const float t = gt[b * n_cat + c];
        if (value_learning) {
          const bool nan = t != t;
          const float mask = nan ? 0.f : 1.f;
          const float d = qb * mask - (nan ? 0.f : t);
          my_loss = 0.5f * d * d;
          g = d * mask * inv_count;
        } else {
          const float d = qb - t;
          my_loss = 0.5f * d * d;
          g = d * inv_count;
        }
      }
    }
    if (dq) dq[i] = from_f32<T>(g);
    if (dq_f32) dq_f32[i] = g;
  }
  __shared__ float red[4];
  float v = my_loss;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_count);
}

// ---------------------------------------------------------------------------------------------------------
// Adam over a flat f32 range (torch.optim.Adam defaults; train_q_network.py:124,227)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float step_size, float beta1, float beta2,
                                                   float omb1, float omb2, float inv_sqrt_bc2, float eps) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define VDQN_ADAM1(c)                                              \
  mm.c = beta1 * mm.c + omb1 * gg.c;                     \
  vv.c = beta2 * vv.c + omb2 * gg.c * gg.c;              \
  pp.c = pp.c - step_size * (mm.c / (sqrtf(vv.c) * inv_sqrt_bc2 + eps));
    VDQN_ADAM1(x) VDQN_ADAM1(y) VDQN_ADAM1(z) VDQN_ADAM1(w)
#undef VDQN_ADAM1
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gg = g[i];
    const float mm = beta1 * m[i] + omb1 * gg;
    const float vv = beta2 * v[i] + omb2 * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
}

inline int grid_for(long total, int cap = 8192) {
  long b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int vdqn_pack_input(const void* src, int32_t src_kind, void* dst, int32_t n_img, int32_t dtype, void* stream) {
  VDQN_CHECK(src && dst && n_img > 0, "vdqn_pack_input: bad args");
  VDQN_CHECK(src_kind == 0 || src_kind == 1, "vdqn_pack_input: src_kind %d", src_kind);
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_pack_input: bad dtype");
  const int g = grid_for((long)n_img * 115 * 115);
  ProfScope ps_("pack_input", 0.0, (double)n_img * (224.0 * 224 * 3 * (src_kind == 0 ? 1 : 4) + 115.0 * 115 * 16 * (dtype == VDQN_BF16 ? 2 : 4)), (hipStream_t)stream);
  if (src_kind == 0 && n_img <= 65535 && (((uintptr_t)src) & 15) == 0) {  // frames are 150528 = 16 * 9408 bytes: every source row is 16-byte aligned
    if (dtype == VDQN_BF16) hipLaunchKernelGGL((pack_input_rows_kernel<bf16raw>), dim3((58 + kPackPairs - 1) / kPackPairs, n_img), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (bf16raw*)dst);
    else hipLaunchKernelGGL((pack_input_rows_kernel<float>), dim3((58 + kPackPairs - 1) / kPackPairs, n_img), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (float*)dst);
    VDQN_LAUNCH_CHECK();
    return VDQN_OK;
  }
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((pack_input_kernel<bf16raw>), dim3(g), dim3(256), 0, (hipStream_t)stream, src, src_kind, (bf16raw*)dst, n_img);
  else hipLaunchKernelGGL((pack_input_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, src, src_kind, (float*)dst, n_img);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_maxpool_fwd(const void* in, void* out, uint8_t* idx, int32_t n_img, int32_t hi, int32_t wi, int32_t c, int32_t dtype,
                                void* stream) {
  VDQN_CHECK(in && out && idx && n_img > 0, "vdqn_maxpool_fwd: bad args");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_maxpool_fwd: bad dtype");
  VDQN_CHECK(c % 8 == 0, "vdqn_maxpool_fwd: channels must be a multiple of 8");
  const int ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
  const int e16 = dtype == VDQN_BF16 ? 8 : 4;
  const int g = grid_for((long)n_img * ho * wo * (c / e16), 65536);
  ProfScope ps_("maxpool_fwd", 0.0, (double)n_img * c * ((double)hi * wi * (16 / e16) + (double)ho * wo * (16 / e16 + 1)), (hipStream_t)stream);
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16raw>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)in, (bf16raw*)out, idx, n_img, hi, wi, c);
  else hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)in, (float*)out, idx, n_img, hi, wi, c);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_maxpool_bwd(const void* gy, const uint8_t* idx, const void* x, void* gx, int32_t n_img, int32_t hi, int32_t wi,
                                int32_t c, int32_t dtype, void* stream) {
  VDQN_CHECK(gy && idx && gx && n_img > 0, "vdqn_maxpool_bwd: bad args");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_maxpool_bwd: bad dtype");
  VDQN_CHECK(c % 8 == 0, "vdqn_maxpool_bwd: channels must be a multiple of 8");
  const int e16 = dtype == VDQN_BF16 ? 8 : 4;
  const int g = grid_for((long)n_img * hi * wi * (c / e16), 65536);
  ProfScope ps_("maxpool_bwd", 0.0, (double)n_img * c * ((x ? 2.0 : 1.0) * hi * wi * (16 / e16) + (double)(hi / 2) * (wi / 2) * (16 / e16 + 1)), (hipStream_t)stream);
  const size_t rows_smem = (size_t)2 * (wi / 2) * c * ((dtype == VDQN_BF16 ? 2 : 4) + 1);
  if (hi % 2 == 0 && wi % 2 == 0 && n_img <= 65535 && rows_smem <= 64 * 1024) {  // row-pair tiles (the 112 x 112 x 64 stem gradient)
    if (dtype == VDQN_BF16)
      hipLaunchKernelGGL((maxpool_bwd_rows_kernel<bf16raw>), dim3(hi / 2, n_img), dim3(256), rows_smem, (hipStream_t)stream, (const bf16raw*)gy, idx, (const bf16raw*)x, (bf16raw*)gx, hi, wi, c);
    else
      hipLaunchKernelGGL((maxpool_bwd_rows_kernel<float>), dim3(hi / 2, n_img), dim3(256), rows_smem, (hipStream_t)stream, (const float*)gy, idx, (const float*)x, (float*)gx, hi, wi, c);
    VDQN_LAUNCH_CHECK();
    return VDQN_OK;
  }
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16raw>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)gy, idx, (const bf16raw*)x, (bf16raw*)gx, n_img, hi, wi, c);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)gy, idx, (const float*)x, (float*)gx, n_img, hi, wi, c);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

__global__ void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int ld, int n_valid) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float m = -INFINITY;
  for (int j = 0; j < n_valid; ++j) m = fmaxf(m, x[(size_t)r * ld + j]);
  float sum = 0.f;
  for (int j = 0; j < n_valid; ++j) sum += expf(x[(size_t)r * ld + j] - m);
  for (int j = 0; j < ld; ++j) y[(size_t)r * ld + j] = j < n_valid ? expf(x[(size_t)r * ld + j] - m) / sum : 0.f;
}

extern "C" int vdqn_softmax_rows(const float* x, float* y, int32_t rows, int32_t ld, int32_t n_valid, void* stream) {
  VDQN_CHECK(x && y && rows > 0 && ld > 0 && n_valid > 0 && n_valid <= ld && n_valid <= 64, "vdqn_softmax_rows: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, x, y, rows, ld, n_valid);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

// mean cross-entropy over rows of logits[rows][ld] (first n_cls columns) against int64 labels, and its gradient
// (softmax - onehot) * inv_count written as T into dlogits[rows][ld] (columns >= n_cls zero): nn.CrossEntropyLoss()
// of train_inverse_model.py:103-104 and its backward
template <typename T>
__global__ void softmax_ce_kernel(const float* __restrict__ x, const int64_t* __restrict__ label, float* __restrict__ loss, T* __restrict__ dx,
                                  int rows, int ld, int n_cls, float inv_count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  float my = 0.f;
  if (r < rows) {
    float m = -INFINITY;
    for (int j = 0; j < n_cls; ++j) m = fmaxf(m, x[(size_t)r * ld + j]);
    float sum = 0.f;
    for (int j = 0; j < n_cls; ++j) sum += expf(x[(size_t)r * ld + j] - m);
    const int lab = (int)label[r];
    my = (logf(sum) + m - x[(size_t)r * ld + lab]) * inv_count;
    for (int j = 0; j < ld; ++j) {
      float g = 0.f;
      if (j < n_cls) g = (expf(x[(size_t)r * ld + j] - m) / sum - (j == lab ? 1.f : 0.f)) * inv_count;
      dx[(size_t)r * ld + j] = from_f32<T>(g);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) my += __shfl_down(my, o, 64);
  if ((threadIdx.x & 63) == 0 && my != 0.f) atomicAdd(loss, my);
}

// out = x * mask * scale (dropout forward and backward with a caller-supplied 0/1 mask of the same dtype)
template <typename T>
__global__ void mask_scale_kernel(const T* __restrict__ x, const T* __restrict__ mask, T* __restrict__ out, long n, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = from_f32<T>(to_f32<T>(x[i]) * to_f32<T>(mask[i]) * scale);
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

extern "C" int vdqn_softmax_ce(const float* logits, const int64_t* labels, float* loss, void* dlogits, int32_t rows, int32_t ld, int32_t n_cls,
                               float inv_count, int32_t dtype, void* stream) {
  VDQN_CHECK(logits && labels && loss && dlogits && rows > 0 && n_cls > 0 && n_cls <= ld, "vdqn_softmax_ce: bad args");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_softmax_ce: bad dtype");
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((softmax_ce_kernel<bf16raw>), dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits, labels, loss, (bf16raw*)dlogits, rows, ld, n_cls, inv_count);
  else hipLaunchKernelGGL((softmax_ce_kernel<float>), dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits, labels, loss, (float*)dlogits, rows, ld, n_cls, inv_count);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_mask_scale(const void* x, const void* mask, void* out, int64_t n, float scale, int32_t dtype, void* stream) {
  VDQN_CHECK(x && mask && out && n > 0, "vdqn_mask_scale: bad args");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_mask_scale: bad dtype");
  const int g = grid_for(n, 4096);
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((mask_scale_kernel<bf16raw>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)x, (const bf16raw*)mask, (bf16raw*)out, (long)n, scale);
  else hipLaunchKernelGGL((mask_scale_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)mask, (float*)out, (long)n, scale);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_axpy(float* y, const float* x, float alpha, int64_t n, void* stream) {
  VDQN_CHECK(y && x && n > 0, "vdqn_axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, (long)n);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_td_loss(const vdqn_td_args* a, void* stream) {
  VDQN_CHECK(a && a->q_before && a->q_after_online && a->q_after_target && a->act && a->rew && a->term && a->loss, "vdqn_td_loss: null arg");
  VDQN_CHECK(!a->use_valid || a->valid, "vdqn_td_loss: use_valid without valid mask");
  VDQN_CHECK(a->batch > 0 && a->n_cat > 0 && a->n_act > 0 && a->ldq >= a->n_cat * a->n_act, "vdqn_td_loss: bad dims");
  VDQN_CHECK(a->dtype == VDQN_F32 || a->dtype == VDQN_BF16, "vdqn_td_loss: bad dtype");
  VDQN_CHECK(a->loss_kind == 0 || a->loss_kind == 1, "vdqn_td_loss: loss_kind %d (0 = half squared error, 1 = Huber)", a->loss_kind);
  const int g = a->deterministic ? 1 : (a->batch * a->ldq + 255) / 256;
  ProfScope ps_("td_loss", 0.0, (double)a->batch * a->ldq * 16.0, (hipStream_t)stream);
  if (a->dtype == VDQN_BF16) hipLaunchKernelGGL((td_loss_kernel<bf16raw>), dim3(g), dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL((td_loss_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, *a);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_gt_loss(const float* q_before, const int64_t* act, const float* gt, float* loss, void* dq, float* dq_f32, int32_t batch,
                            int32_t n_cat, int32_t n_act, int32_t ldq, float inv_count, int32_t value_learning, int32_t dtype, void* stream) {
  VDQN_CHECK(q_before && act && gt && loss, "vdqn_gt_loss: null arg");
  VDQN_CHECK(batch > 0 && ldq >= n_cat * n_act, "vdqn_gt_loss: bad dims");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_gt_loss: bad dtype");
  const int g = (batch * ldq + 255) / 256;
  ProfScope ps_("gt_loss", 0.0, (double)batch * ldq * 8.0, (hipStream_t)stream);
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((gt_loss_kernel<bf16raw>), dim3(g), dim3(256), 0, (hipStream_t)stream, q_before, act, gt, loss, (bf16raw*)dq, dq_f32, batch, n_cat, n_act, ldq, inv_count, value_learning);
  else hipLaunchKernelGGL((gt_loss_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, q_before, act, gt, loss, (float*)dq, dq_f32, batch, n_cat, n_act, ldq, inv_count, value_learning);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_adam(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, double beta1, double beta2, double eps,
                         void* stream) {
  VDQN_CHECK(p && g && m && v && n > 0 && step >= 1, "vdqn_adam: bad args");
  VDQN_CHECK((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "vdqn_adam: pointers must be 16-byte aligned");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int grid = grid_for(n / 4 + 1, 4096);
  ProfScope ps_("adam", 0.0, (double)n * 28.0, (hipStream_t)stream);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, step_size, (float)beta1, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), inv_sqrt_bc2, (float)eps);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
