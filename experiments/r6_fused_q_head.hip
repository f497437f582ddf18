// The Q-head's MLP as ONE launch per pass (round 6; bf16): `top` = Linear(1600 F, 512) - ReLU - Linear(512, 256) - ReLU -
// Linear(256, A * C) forward (archs/HabitatDQNMultiAction.py:31,52-54), and — behind the TD loss — their three data gradients with
// the loss itself in front (train_q_network.py:134-169,180 and loss.backward() :226 through `top`).
//
// Why: between layer4's last convolution and layer4's first data gradient an update runs nine DEPENDENT launches of 7-23 us each on
// a chip that has nothing else to do (profiles/r6_02_update_timeline_live.txt: ~0.37 ms of wall time from features.8 to the first
// layer4 data gradient, 6 % of the update) — the skinny kernels are short already, what remains is launch-to-launch latency and
// eight barriers between layers.  All three layers are ROW-LOCAL (a sample's activations depend on that sample only), so a workgroup
// that owns 16 samples can walk the whole chain with its intermediate tiles in LDS: top.0 -> (l0 tile, 16 x 512) -> top.2 -> (l1
// tile, 16 x 256) -> top.4, and backwards loss -> dQ tile -> g_l1 -> g_l0 -> g_f8.  Weights come straight from L2 into the MFMA's
// first operand (K-contiguous packed rows: a 16-byte load per lane is the lane's eight k values), four 32-deep chunks ahead; the
// N range of a layer is split over the four waves, nothing is summed across waves.  A workgroup streams all three weight matrices
// (1.9 MB at F = 1): 10-19 us at what one CU draws from L2, once per pass instead of three launches + gaps.
//
// Arithmetic: products of bf16 operands accumulated in f32 in ONE running sum per output element in K order (the skinny kernels add
// four interleaved partial sums: results agree to f32 rounding, not bit for bit); every stored activation / gradient is rounded to
// bf16 exactly where the separate launches round it, so the chain sees the same rounded tiles.  Column-sum partials (the bias
// gradients of the layer below) per 16-row tile, of the ROUNDED values, as the separate kernels write them per 32 rows.
#include <stdlib.h>

#include "igemm_common.h"

namespace {

constexpr int kHeadRows = 16;  // samples per workgroup

struct HeadFwdParams {
  const void* x;   // [M][k0] bf16: features.8's output, flattened per sample
  int M, k0;       // k0 = 1600 F
  const void *w0, *w2, *w4;         // packed forward operands [512][k0], [256][512], [64][256] (bf16, K-contiguous)
  const float *b0, *b2, *b4;        // f32 biases (b4: 64 entries, zero behind the real columns)
  void *l0, *l1, *q;                // [M][512], [M][256], [M][64] bf16
  float* qf;                        // [M][64] f32 (unrounded Q)
};

struct HeadBwdParams {
  // loss
  const float *q_before, *q_after_online, *q_after_target;  // f32 [.][64]
  const int64_t* act;
  const float *rew, *term, *valid;
  float* loss;
  float* q_copy;
  int B, n_cat, n_act, k0;
  float gamma, inv_count;
  int clip_rect, linear, use_valid, loss_kind;
  // chain
  const void *wd4, *wd2, *wd0;      // packed data-gradient operands [256][64], [512][256], [k0][512] (bf16, K-contiguous)
  const void *l1, *l0, *f8;         // ReLU masks: the forward activations [B][256], [B][512], [B][k0]
  void *dq, *g_l1, *g_l0, *g_f8;    // bf16 [B][64], [B][256], [B][512], [B][k0]
  float *p_l1, *p_l0, *p_f8;        // per-16-row-tile column sums [tiles][256], [tiles][512], [tiles][k0]
};

constexpr int kPitch0 = 512 * 2 + 16;  // LDS row pitch of the 512-wide tile: +4 banks per row (a ds_read_b128 group covers 64 banks once)
constexpr int kPitch1 = 256 * 2 + 16;
constexpr int kPitchQ = 64 * 2 + 16;

__device__ __forceinline__ uint4 ld16(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }

// acc[j] += W[n0 + j * 16 + i16][k] * A[i16][k] over k in [0, K): A rows from `a_row` (this lane's row, + g * 16 already applied;
// global or LDS), W rows from global, DEPTH chunks of 32 in flight.  NJ fragments per wave.
template <int NJ, int DEPTH>
__device__ __forceinline__ void gemm_rows(f32x4 (&acc)[NJ], const unsigned char* a_row, const unsigned char* w_row0, long w_row_stride, int n_chunks) {
  uint4 fa[DEPTH], fb[DEPTH][NJ];
  auto load = [&](int d, int c) {
    c = c < n_chunks ? c : n_chunks - 1;  // beyond the end: re-load the last chunk (unused)
    fa[d] = ld16(a_row + (long)c * 64);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[d][j] = ld16(w_row0 + j * w_row_stride + (long)c * 64);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d, d);
  for (int c = 0; c < n_chunks; c += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (c + d < n_chunks) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[d][j]), __builtin_bit_cast(bf16x8, fa[d]), acc[j], 0, 0, 0);
        load(d, c + d + DEPTH);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward: x -> l0 -> l1 -> q.  Lane (i16, g) of a wave ends with columns j * 16 + 4 g .. + 3 of fragment j for row i16.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadFwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char sL0[kHeadRows * kPitch0];
  __shared__ __attribute__((aligned(16))) unsigned char sL1[kHeadRows * kPitch1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int m0 = (int)blockIdx.x * kHeadRows;
  const int m = m0 + i16;
  const bool row_ok = m < p.M;
  const int mc = row_ok ? m : p.M - 1;  // rows past M: clamped loads, nothing stored

  // ---- top.0: 512 columns, 128 per wave ----
  {
    constexpr int NJ = 8;
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* a_row = reinterpret_cast<const unsigned char*>(p.x) + (long)mc * p.k0 * 2 + g * 16;
    const unsigned char* w_row0 = reinterpret_cast<const unsigned char*>(p.w0) + (long)(wave * 128 + i16) * p.k0 * 2 + g * 16;
    gemm_rows<NJ, 4>(acc, a_row, w_row0, (long)16 * p.k0 * 2, p.k0 / 32);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = wave * 128 + j * 16 + g * 4;
      const float4 b = *reinterpret_cast<const float4*>(p.b0 + col);
      bf16raw o[4];
      o[0] = f32_to_bf16(fmaxf(acc[j][0] + b.x, 0.f));
      o[1] = f32_to_bf16(fmaxf(acc[j][1] + b.y, 0.f));
      o[2] = f32_to_bf16(fmaxf(acc[j][2] + b.z, 0.f));
      o[3] = f32_to_bf16(fmaxf(acc[j][3] + b.w, 0.f));
      *reinterpret_cast<uint2*>(sL0 + i16 * kPitch0 + col * 2) = *reinterpret_cast<const uint2*>(o);
      if (row_ok) *reinterpret_cast<uint2*>(reinterpret_cast<bf16raw*>(p.l0) + (size_t)m * 512 + col) = *reinterpret_cast<const uint2*>(o);
    }
  }
  __syncthreads();
  // ---- top.2: 256 columns, 64 per wave; A = the l0 tile in LDS ----
  {
    constexpr int NJ = 4;
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* w_row0 = reinterpret_cast<const unsigned char*>(p.w2) + (long)(wave * 64 + i16) * 512 * 2 + g * 16;
    gemm_rows<NJ, 4>(acc, sL0 + i16 * kPitch0 + g * 16, w_row0, (long)16 * 512 * 2, 512 / 32);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = wave * 64 + j * 16 + g * 4;
      const float4 b = *reinterpret_cast<const float4*>(p.b2 + col);
      bf16raw o[4];
      o[0] = f32_to_bf16(fmaxf(acc[j][0] + b.x, 0.f));
      o[1] = f32_to_bf16(fmaxf(acc[j][1] + b.y, 0.f));
      o[2] = f32_to_bf16(fmaxf(acc[j][2] + b.z, 0.f));
      o[3] = f32_to_bf16(fmaxf(acc[j][3] + b.w, 0.f));
      *reinterpret_cast<uint2*>(sL1 + i16 * kPitch1 + col * 2) = *reinterpret_cast<const uint2*>(o);
      if (row_ok) *reinterpret_cast<uint2*>(reinterpret_cast<bf16raw*>(p.l1) + (size_t)m * 256 + col) = *reinterpret_cast<const uint2*>(o);
    }
  }
  __syncthreads();
  // ---- top.4: 64 (padded) columns, 16 per wave; no ReLU; the unrounded f32 copy is what the loss reads ----
  {
    f32x4 acc[1];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* w_row0 = reinterpret_cast<const unsigned char*>(p.w4) + (long)(wave * 16 + i16) * 256 * 2 + g * 16;
    gemm_rows<1, 4>(acc, sL1 + i16 * kPitch1 + g * 16, w_row0, 0, 256 / 32);
    const int col = wave * 16 + g * 4;
    const float4 b = *reinterpret_cast<const float4*>(p.b4 + col);
    const float x0 = acc[0][0] + b.x, x1 = acc[0][1] + b.y, x2 = acc[0][2] + b.z, x3 = acc[0][3] + b.w;
    if (row_ok) {
      bf16raw o[4] = {f32_to_bf16(x0), f32_to_bf16(x1), f32_to_bf16(x2), f32_to_bf16(x3)};
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16raw*>(p.q) + (size_t)m * 64 + col) = *reinterpret_cast<const uint2*>(o);
      *reinterpret_cast<float4*>(p.qf + (size_t)m * 64 + col) = make_float4(x0, x1, x2, x3);
    }
  }
}

// sum over the 16 rows (lanes i16 = 0..15 of one g group) of the four values a lane holds: xor-shuffles inside the 16-lane group
__device__ __forceinline__ f32x4 colsum16(f32x4 v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    v[0] += __shfl_xor(v[0], o, 64);
    v[1] += __shfl_xor(v[1], o, 64);
    v[2] += __shfl_xor(v[2], o, 64);
    v[3] += __shfl_xor(v[3], o, 64);
  }
  return v;
}

// epilogue of one data-gradient fragment: ReLU mask of the layer's input activation, bf16 rounding, tile (LDS, optional) + global
// store, column sum of the rounded values over the tile's valid rows
__device__ __forceinline__ void dgrad_frag_out(const f32x4 acc, const bf16raw* __restrict__ mask_row, bf16raw* __restrict__ out_row, unsigned char* lds_row,
                                               float* __restrict__ part_row, int col, bool row_ok, int i16) {
  float x[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (row_ok) {
    const uint2 mv = *reinterpret_cast<const uint2*>(mask_row + col);
    const bf16raw* pm = reinterpret_cast<const bf16raw*>(&mv);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = (bf16_to_f32(pm[e]) > 0.f) ? x[e] : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = 0.f;
  }
  bf16raw o[4];
  f32x4 back;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = f32_to_bf16(x[e]);
    back[e] = bf16_to_f32(o[e]);
  }
  if (lds_row) *reinterpret_cast<uint2*>(lds_row + col * 2) = *reinterpret_cast<const uint2*>(o);
  if (row_ok) *reinterpret_cast<uint2*>(out_row + col) = *reinterpret_cast<const uint2*>(o);
  const f32x4 s = colsum16(back);
  if (i16 == 0) *reinterpret_cast<float4*>(part_row + col) = make_float4(s[0], s[1], s[2], s[3]);
}

// ---------------------------------------------------------------------------------------------------------
// backward: TD loss -> dQ -> g_l1 -> g_l0 -> g_f8 for the workgroup's 16 samples
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_kernel(const HeadBwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char sDq[kHeadRows * kPitchQ];
  __shared__ __attribute__((aligned(16))) unsigned char sG1[kHeadRows * kPitch1];
  __shared__ __attribute__((aligned(16))) unsigned char sG0[kHeadRows * kPitch0];
  __shared__ float sRed[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int tile = (int)blockIdx.x;
  const int m0 = tile * kHeadRows;
  const int m = m0 + i16;
  const bool row_ok = m < p.B;

  // ---- the TD loss of these 16 samples (the arithmetic of td_loss_kernel, pointwise.hip): one element per thread and pass ----
  {
    const int nq = p.n_cat * p.n_act;
    float my_loss = 0.f;
#pragma unroll
    for (int pass = 0; pass < kHeadRows * 64 / 256; ++pass) {
      const int i = pass * 256 + tid;
      const int r = i >> 6, col = i & 63;
      const int b = m0 + r;
      float gq = 0.f;
      if (b < p.B && col < nq) {
        const int c = col / p.n_act, ac = col - c * p.n_act;
        const int act = (int)p.act[b];
        if (p.q_copy) p.q_copy[(size_t)b * nq + col] = p.q_before[(size_t)b * 64 + col];
        if (ac == act) {
          const float qb = p.q_before[(size_t)b * 64 + col];
          const float* qo = p.q_after_online + (size_t)b * 64 + c * p.n_act;
          int best = 0;
          float bv = qo[0];
          for (int k = 1; k < p.n_act; ++k) {
            const float v = qo[k];
            if (v > bv) {  // strict: first maximum wins (torch.argmax)
              bv = v;
              best = k;
            }
          }
          float qa = p.q_after_target[(size_t)b * 64 + c * p.n_act + best];
          qa = qa * (1.0f - p.term[b * p.n_cat + c]);
          const float rw = p.rew[b * p.n_cat + c];
          float y = p.linear ? rw + (qa - 0.1f) : rw + p.gamma * qa;
          if (p.clip_rect) y = fminf(fmaxf(y, 0.f), 1.f);
          const float d = qb - y;
          const float vm = p.use_valid ? p.valid[b * p.n_cat + c] : 1.0f;
          if (p.loss_kind == 1) {
            const float ad = fabsf(d);
            my_loss += (ad < 1.0f ? 0.5f * d * d : ad - 0.5f) * vm;
            gq = fminf(fmaxf(d, -1.0f), 1.0f) * vm * p.inv_count;
          } else {
            my_loss += 0.5f * d * d * vm;
            gq = d * vm * p.inv_count;
          }
        }
      }
      const bf16raw o = f32_to_bf16(gq);
      *reinterpret_cast<bf16raw*>(sDq + r * kPitchQ + col * 2) = o;
      if (b < p.B) reinterpret_cast<bf16raw*>(p.dq)[(size_t)b * 64 + col] = o;
    }
    float v = my_loss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) sRed[wave] = v;
    __syncthreads();  // (also: the dQ tile is complete)
    if (tid == 0) {
      const float s = sRed[0] + sRed[1] + sRed[2] + sRed[3];
      if (s != 0.f) atomicAdd(p.loss, s * p.inv_count);
    }
  }
  // ---- top.4's data gradient: g_l1 [16][256] = dQ [16][64] . Wd4 [256][64]^T, masked by l1 > 0; 64 columns per wave ----
  {
    constexpr int NJ = 4;
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* w_row0 = reinterpret_cast<const unsigned char*>(p.wd4) + (long)(wave * 64 + i16) * 64 * 2 + g * 16;
    gemm_rows<NJ, 2>(acc, sDq + i16 * kPitchQ + g * 16, w_row0, (long)16 * 64 * 2, 64 / 32);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      dgrad_frag_out(acc[j], reinterpret_cast<const bf16raw*>(p.l1) + (size_t)(row_ok ? m : 0) * 256, reinterpret_cast<bf16raw*>(p.g_l1) + (size_t)(row_ok ? m : 0) * 256,
                     sG1 + i16 * kPitch1, p.p_l1 + (size_t)tile * 256, wave * 64 + j * 16 + g * 4, row_ok, i16);
  }
  __syncthreads();
  // ---- top.2's data gradient: g_l0 [16][512] = g_l1 . Wd2 [512][256]^T, masked by l0 > 0; 128 columns per wave ----
  {
    constexpr int NJ = 8;
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* w_row0 = reinterpret_cast<const unsigned char*>(p.wd2) + (long)(wave * 128 + i16) * 256 * 2 + g * 16;
    gemm_rows<NJ, 4>(acc, sG1 + i16 * kPitch1 + g * 16, w_row0, (long)16 * 256 * 2, 256 / 32);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      dgrad_frag_out(acc[j], reinterpret_cast<const bf16raw*>(p.l0) + (size_t)(row_ok ? m : 0) * 512, reinterpret_cast<bf16raw*>(p.g_l0) + (size_t)(row_ok ? m : 0) * 512,
                     sG0 + i16 * kPitch0, p.p_l0 + (size_t)tile * 512, wave * 128 + j * 16 + g * 4, row_ok, i16);
  }
  __syncthreads();
  // ---- top.0's data gradient: g_f8 [16][k0] = g_l0 . Wd0 [k0][512]^T, masked by f8 > 0; 16-column fragments dealt round-robin to the
  // waves, five at a time ----
  {
    constexpr int NJ = 5;
    const int n_frag = p.k0 / 16;  // 100 F
    for (int f0 = wave * NJ; f0 < n_frag; f0 += 4 * NJ) {
      f32x4 acc[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // (fragments past the end of a ragged last group re-read the last fragment's rows; their results are dropped)
      const int fl = n_frag - 1;
      uint4 fa[2], fb[2][NJ];
      const unsigned char* a_row = sG0 + i16 * kPitch0 + g * 16;
      const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.wd0) + (long)i16 * 512 * 2 + g * 16;
      long woff[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) woff[j] = (long)((f0 + j <= fl ? f0 + j : fl) * 16) * 512 * 2;
      auto load = [&](int d, int c) {
        c = c < 16 ? c : 15;
        fa[d] = ld16(a_row + c * 64);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[d][j] = ld16(wb + woff[j] + c * 64);
      };
      load(0, 0);
      load(1, 1);
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[d][j]), __builtin_bit_cast(bf16x8, fa[d]), acc[j], 0, 0, 0);
          load(d, c + d + 2);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (f0 + j > fl) continue;  // (wave-uniform)
        dgrad_frag_out(acc[j], reinterpret_cast<const bf16raw*>(p.f8) + (size_t)(row_ok ? m : 0) * p.k0, reinterpret_cast<bf16raw*>(p.g_f8) + (size_t)(row_ok ? m : 0) * p.k0,
                       nullptr, p.p_f8 + (size_t)tile * p.k0, (f0 + j) * 16 + g * 4, row_ok, i16);
      }
    }
  }
}

}  // namespace

// measurement / test hook (not part of include/vdqn.h): 0 = the Q-head on the separate per-layer launches (rounds 4-5), -1 / 1 = fused
static int g_head_fused = -1;
extern "C" void vdqn_debug_set_head_fused(int v) { g_head_fused = v; }
bool vdqn_head_fused_enabled() {
  static const bool env_on = [] { const char* e = getenv("VDQN_HEAD_FUSED"); return !(e && e[0] == '0'); }();
  return g_head_fused < 0 ? env_on : g_head_fused != 0;
}
int vdqn_head_part_rows() { return kHeadRows; }

// x [M][k0] -> l0 [M][512], l1 [M][256], q [M][64] (bf16), qf [M][64] (f32); packed bf16 weights and f32 biases of top.0 / top.2 / top.4
int vdqn_launch_head_fwd(const void* x, int M, int k0, const void* w0, const float* b0, const void* w2, const float* b2, const void* w4, const float* b4,
                         void* l0, void* l1, void* q, float* qf, hipStream_t stream) {
  VDQN_CHECK(x && w0 && w2 && w4 && b0 && b2 && b4 && l0 && l1 && q && qf && M > 0 && k0 > 0 && k0 % 32 == 0, "head forward: bad arguments");
  HeadFwdParams p;
  p.x = x; p.M = M; p.k0 = k0;
  p.w0 = w0; p.w2 = w2; p.w4 = w4; p.b0 = b0; p.b2 = b2; p.b4 = b4;
  p.l0 = l0; p.l1 = l1; p.q = q; p.qf = qf;
  const double flops = 2.0 * M * ((double)k0 * 512 + 512.0 * 256 + 256.0 * 64);
  const double bytes = 2.0 * ((double)M * (k0 + 512 + 256 + 64) + (double)k0 * 512 + 512.0 * 256 + 256.0 * 64);
  ProfScope ps_("head_mlp<bf16,fwd>", flops, bytes, stream);
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((M + kHeadRows - 1) / kHeadRows)), dim3(256), 0, stream, p);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

// TD loss + dQ + the three data gradients of `top` for B samples; t: the loss arguments (q_*, act, rew, term, valid, loss, q_copy, dq as the
// bf16 dQ buffer; dtype must be bf16, deterministic 0); part_*: per-16-row-tile column sums
int vdqn_launch_head_bwd(const vdqn_td_args* t, int k0, const void* wd4, const void* wd2, const void* wd0, const void* l1, const void* l0, const void* f8,
                         void* g_l1, void* g_l0, void* g_f8, float* p_l1, float* p_l0, float* p_f8, hipStream_t stream) {
  VDQN_CHECK(t && t->q_before && t->q_after_online && t->q_after_target && t->act && t->rew && t->term && t->loss && t->dq, "head backward: null loss argument");
  VDQN_CHECK(t->ldq == 64 && t->dtype == VDQN_BF16 && !t->deterministic && !t->dq_f32 && k0 > 0 && k0 % 16 == 0, "head backward: unsupported loss arguments");
  VDQN_CHECK(wd4 && wd2 && wd0 && l1 && l0 && f8 && g_l1 && g_l0 && g_f8 && p_l1 && p_l0 && p_f8, "head backward: null chain argument");
  HeadBwdParams p;
  p.q_before = t->q_before; p.q_after_online = t->q_after_online; p.q_after_target = t->q_after_target;
  p.act = t->act; p.rew = t->rew; p.term = t->term; p.valid = t->valid; p.loss = t->loss; p.q_copy = t->q_copy;
  p.B = t->batch; p.n_cat = t->n_cat; p.n_act = t->n_act; p.k0 = k0;
  p.gamma = t->gamma; p.inv_count = t->inv_count;
  p.clip_rect = t->clip_rect; p.linear = t->linear; p.use_valid = t->use_valid; p.loss_kind = t->loss_kind;
  p.wd4 = wd4; p.wd2 = wd2; p.wd0 = wd0; p.l1 = l1; p.l0 = l0; p.f8 = f8;
  p.dq = t->dq; p.g_l1 = g_l1; p.g_l0 = g_l0; p.g_f8 = g_f8; p.p_l1 = p_l1; p.p_l0 = p_l0; p.p_f8 = p_f8;
  const double flops = 2.0 * t->batch * (64.0 * 256 + 256.0 * 512 + 512.0 * k0);
  const double bytes = 2.0 * ((double)t->batch * 2 * (64 + 256 + 512 + k0) + 64.0 * 256 + 256.0 * 512 + 512.0 * k0);
  ProfScope ps_("head_mlp<bf16,loss+dgrad>", flops, bytes, stream);
  hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((t->batch + kHeadRows - 1) / kHeadRows)), dim3(256), 0, stream, p);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
