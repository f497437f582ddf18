// Train-mode BatchNorm (batch statistics) and the 7x7 average pool of ARCHITECTURE='basic'
// (archs/HabitatDQNMultiAction.py:32-34,37-40: without extra_capacity the whole ResNet stays in train mode, so all 20
// BatchNorm layers normalise with the statistics of the current minibatch and update their running statistics on
// every online forward; torch.nn.BatchNorm2d semantics: biased variance for normalisation, unbiased for the running
// estimate, momentum 0.1, eps 1e-5).  HBM-bound passes over the NHWC conv output y[M][C] (16-byte vectors):
//
//   forward : sums = (sum y, sum y^2)  ->  mean, rstd, scale = gamma*rstd, shift = beta - mean*scale, running stats
//             z = relu?(y*scale + shift (+ residual))
//   backward: sums = (sum g, sum g*xhat)  ->  dbeta, dgamma,  dy = scale * (g - sum_g/M - xhat * sum_gxhat/M)
//
// Statistic groups: the reference applies `features` to one frame slot at a time (archs/...:49-51) and calls the model
// once on `before` and once on `after` (train_q_network.py:131,142), so with F frames per sample and the engine's
// single pass over [before; after] there are G = 2F independent minibatches: image i (sample-major, frame-minor)
// belongs to group (i / imgs_per_half) * F + i % F.  The running statistics are updated group after group in the
// reference's call order (before f=0..F-1, after f=0..F-1).
//
// work[G][6][C] f32 per BatchNorm layer: mean, rstd, scale, shift, sumA, sumB.
#include "common.h"

namespace {

struct BnGeom {
  int hw;          // pixels per image
  int frames;      // F
  int iph;         // images per half (= samples per call * F)
  int groups;      // G
  int grp_rows;    // rows per group = (iph / F) * hw
};

// grid (blocks, G).  MODE 0: (sum y, sum y^2)   MODE 1: (sum g, sum g * xhat)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_sums_kernel(const T* __restrict__ a, const T* __restrict__ y, float* __restrict__ work, BnGeom gm, int C,
                                                      int rows_per_block, float* __restrict__ det_part) {
  constexpr int E16 = 16 / (int)sizeof(T);
  __shared__ float red[2][256 * E16];
  const int grp = blockIdx.y;
  float* wk = work + (size_t)grp * 6 * C;
  const int half = grp / gm.frames, f = grp - half * gm.frames;
  const int cg_n = C / E16, nstripe = 256 / cg_n;
  const int cg = threadIdx.x % cg_n, stripe = threadIdx.x / cg_n;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(gm.grp_rows, r0 + rows_per_block);
  float s1[E16], s2[E16], mu[E16], rs[E16];
#pragma unroll
  for (int e = 0; e < E16; ++e) {
    s1[e] = s2[e] = 0.f;
    mu[e] = MODE == 1 ? wk[cg * E16 + e] : 0.f;
    rs[e] = MODE == 1 ? wk[C + cg * E16 + e] : 0.f;
  }
  if (stripe < nstripe) {
    const int smp_half = gm.iph / gm.frames;
    // row of the j-th pixel of this group (contiguous when there is one frame slot)
    auto row_of = [&](int j) -> size_t {
      if (gm.frames == 1) return (size_t)half * gm.grp_rows + j;
      const int smp = j / gm.hw, p = j - smp * gm.hw;
      return ((size_t)(half * smp_half + smp) * gm.frames + f) * gm.hw + p;
    };
    auto accum = [&](const uint4& va, const uint4& vy) {
      const T* pa = reinterpret_cast<const T*>(&va);
      const T* py = reinterpret_cast<const T*>(&vy);
#pragma unroll
      for (int e = 0; e < E16; ++e) {
        const float v = to_f32<T>(pa[e]);
        s1[e] += v;
        s2[e] += MODE == 0 ? v * v : v * (to_f32<T>(py[e]) - mu[e]) * rs[e];
      }
    };
    constexpr int U = 4;  // independent 16-byte loads in flight per thread
    int j = r0 + stripe;
    for (; j + (U - 1) * nstripe < r1; j += U * nstripe) {
      uint4 va[U], vy[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t row = row_of(j + u * nstripe);
        va[u] = *reinterpret_cast<const uint4*>(a + row * C + cg * E16);
        if (MODE == 1) vy[u] = *reinterpret_cast<const uint4*>(y + row * C + cg * E16);
        else vy[u] = va[u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accum(va[u], vy[u]);
    }
    for (; j < r1; j += nstripe) {
      const size_t row = row_of(j);
      const uint4 va = *reinterpret_cast<const uint4*>(a + row * C + cg * E16);
      const uint4 vy = MODE == 1 ? *reinterpret_cast<const uint4*>(y + row * C + cg * E16) : va;
      accum(va, vy);
    }
  }
#pragma unroll
  for (int e = 0; e < E16; ++e) {
    red[0][threadIdx.x * E16 + e] = s1[e];
    red[1][threadIdx.x * E16 + e] = s2[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / E16, e = c % E16;
    float t1 = 0.f, t2 = 0.f;
    for (int st = 0; st < nstripe; ++st) {
      t1 += red[0][(st * cg_n + g) * E16 + e];
      t2 += red[1][(st * cg_n + g) * E16 + e];
    }
    if (det_part) {  // deterministic mode: this block's partial sums as plain stores, added up in block order by bn_sums_reduce_kernel
      float* o = det_part + ((size_t)grp * gridDim.x + blockIdx.x) * 2 * C;
      o[c] = t1;
      o[C + c] = t2;
    } else {
      atomicAdd(wk + 4 * C + c, t1);
      atomicAdd(wk + 5 * C + c, t2);
    }
  }
}

// deterministic mode, second stage: work[g][4..5][c] = sum over the blocks of group g of their partial sums, in block order
__global__ void bn_sums_reduce_kernel(const float* __restrict__ part, float* __restrict__ work, int blocks, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * C) return;
  const int grp = blockIdx.y;
  const float* p = part + (size_t)grp * blocks * 2 * C + i;
  float t = 0.f;
  for (int b = 0; b < blocks; ++b) t += p[(size_t)b * 2 * C];
  work[(size_t)grp * 6 * C + 4 * C + i] = t;
}

// one thread per channel; the groups are consumed in order so the running statistics see the reference's update order
__global__ void bn_finalize_kernel(float* __restrict__ work, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int groups, int M, int C, float momentum,
                                   float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv = 1.0f / (float)M;
  const float ga = gamma[c], be = beta[c];
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  for (int g = 0; g < groups; ++g) {
    float* wk = work + (size_t)g * 6 * C;
    const float mean = wk[4 * C + c] * inv;
    float var = wk[5 * C + c] * inv - mean * mean;
    var = fmaxf(var, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float scale = ga * rstd;
    wk[c] = mean;
    wk[C + c] = rstd;
    wk[2 * C + c] = scale;
    wk[3 * C + c] = be - mean * scale;
    wk[4 * C + c] = 0.f;  // leave the sums zeroed for the backward pass
    wk[5 * C + c] = 0.f;
    const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    rm = (1.0f - momentum) * rm + momentum * mean;
    rv = (1.0f - momentum) * rv + momentum * unbiased;
  }
  if (running_mean) {
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

// grid (chunks, images): every block works inside one image, so the statistic group is a per-block constant
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ resid, T* __restrict__ z,
                                                       const float* __restrict__ work, BnGeom gm, int C, int relu) {
  constexpr int E16 = 16 / (int)sizeof(T);
  const int img = blockIdx.y;
  const int grp = (img / gm.iph) * gm.frames + img % gm.frames;
  const float* wk = work + (size_t)grp * 6 * C;
  const int cg_mask = C / E16 - 1;  // C / E16 is a power of two for every ResNet-18 width
  const int per_img = gm.hw * (C / E16);
  const size_t base = (size_t)img * per_img;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < per_img; v += gridDim.x * blockDim.x) {
    const int cg = v & cg_mask;
    const uint4 vy = reinterpret_cast<const uint4*>(y)[base + v];
    const T* py = reinterpret_cast<const T*>(&vy);
    uint4 vr = make_uint4(0, 0, 0, 0);
    if (resid) vr = reinterpret_cast<const uint4*>(resid)[base + v];
    const T* pr = reinterpret_cast<const T*>(&vr);
    T o[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) {
      const int c = cg * E16 + e;
      float t = to_f32<T>(py[e]) * wk[2 * C + c] + wk[3 * C + c];
      if (resid) t += to_f32<T>(pr[e]);
      if (relu) t = fmaxf(t, 0.f);
      o[e] = from_f32<T>(t);
    }
    reinterpret_cast<uint4*>(z)[base + v] = *reinterpret_cast<const uint4*>(o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ dy,
                                                           const float* __restrict__ work, BnGeom gm, int C, float inv_m) {
  constexpr int E16 = 16 / (int)sizeof(T);
  const int img = blockIdx.y;
  const int grp = (img / gm.iph) * gm.frames + img % gm.frames;
  const float* wk = work + (size_t)grp * 6 * C;
  const int cg_mask = C / E16 - 1;
  const int per_img = gm.hw * (C / E16);
  const size_t base = (size_t)img * per_img;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < per_img; v += gridDim.x * blockDim.x) {
    const int cg = v & cg_mask;
    const uint4 vg = reinterpret_cast<const uint4*>(g)[base + v];
    const uint4 vy = reinterpret_cast<const uint4*>(y)[base + v];
    const T* pg = reinterpret_cast<const T*>(&vg);
    const T* py = reinterpret_cast<const T*>(&vy);
    T o[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) {
      const int c = cg * E16 + e;
      const float xhat = (to_f32<T>(py[e]) - wk[c]) * wk[C + c];
      o[e] = from_f32<T>(wk[2 * C + c] * (to_f32<T>(pg[e]) - wk[4 * C + c] * inv_m - xhat * wk[5 * C + c] * inv_m));
    }
    reinterpret_cast<uint4*>(dy)[base + v] = *reinterpret_cast<const uint4*>(o);
  }
}

// dgamma = sum over groups of sum(g * xhat), dbeta = sum over groups of sum(g); re-zeroes nothing (the forward does)
__global__ void bn_param_grad_kernel(const float* __restrict__ work, float* __restrict__ dgamma, float* __restrict__ dbeta, int groups, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int g = 0; g < groups; ++g) {
    a += work[(size_t)g * 6 * C + 5 * C + c];
    b += work[(size_t)g * 6 * C + 4 * C + c];
  }
  if (dgamma) dgamma[c] = a;
  if (dbeta) dbeta[c] = b;
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int n, int hw, int C) {
  constexpr int E16 = 16 / (int)sizeof(T);
  const int cg_n = C / E16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * cg_n) return;
  const int img = i / cg_n, cg = i - img * cg_n;
  float s[E16];
#pragma unroll
  for (int e = 0; e < E16; ++e) s[e] = 0.f;
  for (int p = 0; p < hw; ++p) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + ((size_t)img * hw + p) * C + cg * E16);
    const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int e = 0; e < E16; ++e) s[e] += to_f32<T>(pv[e]);
  }
  T o[E16];
  const float inv = 1.0f / (float)hw;
#pragma unroll
  for (int e = 0; e < E16; ++e) o[e] = from_f32<T>(s[e] * inv);
  *reinterpret_cast<uint4*>(out + (size_t)img * C + cg * E16) = *reinterpret_cast<const uint4*>(o);
}

// gx[n][p][c] = (x[n][p][c] > 0) * g[n][c] / hw     (average pool + the ReLU that produced x)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x, T* __restrict__ gx, int n, int hw, int C) {
  constexpr int E16 = 16 / (int)sizeof(T);
  const int cg_n = C / E16;
  const long total = (long)n * hw * cg_n;
  const float inv = 1.0f / (float)hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_n);
    const long pix = i / cg_n;
    const int img = (int)(pix / hw);
    const uint4 vg = *reinterpret_cast<const uint4*>(g + (size_t)img * C + cg * E16);
    const uint4 vx = reinterpret_cast<const uint4*>(x)[i];
    const T* pg = reinterpret_cast<const T*>(&vg);
    const T* px = reinterpret_cast<const T*>(&vx);
    T o[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) o[e] = from_f32<T>(to_f32<T>(px[e]) > 0.f ? to_f32<T>(pg[e]) * inv : 0.f);
    reinterpret_cast<uint4*>(gx)[i] = *reinterpret_cast<const uint4*>(o);
  }
}

// cross-rank statistics (SyncBN): the two sums of every group are packed into one contiguous buffer, all-reduced by
// the host's callback, and unpacked again.  dir = 0: work -> buf, 1: buf -> work
__global__ void bn_sync_pack_kernel(float* __restrict__ work, float* __restrict__ buf, int groups, int C, int dir) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * 2 * C) return;
  const int g = i / (2 * C), r = i - g * 2 * C;
  float* w = work + (size_t)g * 6 * C + 4 * C + r;
  if (dir == 0) buf[i] = *w;
  else *w = buf[i];
}

inline int grid_rows(int M, int* rpb) {
  int blocks = (M + 511) / 512;  // measured: more, smaller blocks lose to the per-block reduction + atomics
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  *rpb = (M + blocks - 1) / blocks;
  return blocks;
}
inline int grid_elems(long total) {
  long b = (total + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}
inline int apply_chunks(int per_img) {
  int b = (per_img + 1023) / 1024;  // about four 16-byte vectors per thread
  return b < 1 ? 1 : b;
}

template <typename T, int MODE>
void launch_bn_sums(const void* a, const void* y, float* work, const BnGeom& gm, int c, int blocks, int rpb, float* det_ws, hipStream_t st) {
  hipLaunchKernelGGL((bn_sums_kernel<T, MODE>), dim3(blocks, gm.groups), dim3(256), 0, st, (const T*)a, (const T*)y, work, gm, c, rpb, det_ws);
  if (det_ws) hipLaunchKernelGGL(bn_sums_reduce_kernel, dim3((2 * c + 255) / 256, gm.groups), dim3(256), 0, st, det_ws, work, blocks, c);
}

int make_geom(const char* who, int n_img, int hw, int c, int frames, int iph, int dtype, BnGeom* gm) {
  VDQN_CHECK(n_img > 0 && hw > 0 && c > 0 && frames > 0 && iph > 0, "%s: bad sizes", who);
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "%s: bad dtype", who);
  VDQN_CHECK(iph % frames == 0 && n_img % iph == 0, "%s: n_img %d / imgs_per_half %d / frames %d do not divide", who, n_img, iph, frames);
  const int e16 = dtype == VDQN_BF16 ? 8 : 4;
  const int cg = c / e16;
  VDQN_CHECK(c % e16 == 0 && cg <= 256 && (cg & (cg - 1)) == 0, "%s: channel count %d unsupported", who, c);
  VDQN_CHECK(n_img <= 65535, "%s: more than 65535 images per call", who);
  gm->hw = hw;
  gm->frames = frames;
  gm->iph = iph;
  gm->groups = n_img / iph * frames;
  gm->grp_rows = iph / frames * hw;
  return VDQN_OK;
}

// deterministic mode: the workspace that takes one partial sum pair per block (nullptr = atomics)
int det_workspace(const char* who, const BnSync* sync, const BnGeom& gm, int blocks, int c, float** out) {
  *out = nullptr;
  if (!sync || !sync->det_ws) return VDQN_OK;
  const int64_t need = (int64_t)gm.groups * blocks * 2 * c * 4;
  VDQN_CHECK(sync->det_ws_bytes >= need, "%s: deterministic workspace of %lld bytes, %lld needed", who, (long long)sync->det_ws_bytes, (long long)need);
  VDQN_CHECK(((uintptr_t)sync->det_ws & 15) == 0, "%s: workspace must be 16-byte aligned", who);
  *out = sync->det_ws;
  return VDQN_OK;
}

}  // namespace

extern "C" int64_t vdqn_bn_train_workspace_bytes(int32_t n_img, int32_t hw, int32_t c, int32_t num_frames, int32_t imgs_per_half) {
  BnGeom gm;
  if (make_geom("vdqn_bn_train_workspace_bytes", n_img, hw, c, num_frames, imgs_per_half, VDQN_F32, &gm) != VDQN_OK) return -1;
  int rpb;
  return (int64_t)gm.groups * grid_rows(gm.grp_rows, &rpb) * 2 * c * 4;
}

// sync (may be null): all-reduce hook + scratch for SyncBN; the statistics then cover grp_rows * sync->world rows
int vdqn_bn_train_fwd_impl(const void* y, const void* resid, void* z, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* work, int32_t n_img, int32_t hw, int32_t c, int32_t num_frames,
                           int32_t imgs_per_half, int32_t relu, float momentum, float eps, int32_t dtype, void* stream, const BnSync* sync) {
  VDQN_CHECK(y && z && gamma && beta && work, "vdqn_bn_train_fwd: null arg");
  BnGeom gm;
  if (int rc = make_geom("vdqn_bn_train_fwd", n_img, hw, c, num_frames, imgs_per_half, dtype, &gm)) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(work, 0, (size_t)gm.groups * 6 * c * sizeof(float), st);
  VDQN_CHECK(e == hipSuccess, "vdqn_bn_train_fwd: memset failed: %s", hipGetErrorString(e));
  int rpb;
  const int blocks = grid_rows(gm.grp_rows, &rpb);
  float* det_ws = nullptr;
  if (int rc = det_workspace("vdqn_bn_train_fwd", sync, gm, blocks, c, &det_ws)) return rc;
  const double esz = dtype == VDQN_BF16 ? 2 : 4, elems = (double)n_img * hw * c;
  {
    ProfScope ps("bn_stats", 0.0, elems * esz, st);
    if (dtype == VDQN_BF16) launch_bn_sums<bf16raw, 0>(y, nullptr, work, gm, c, blocks, rpb, det_ws, st);
    else launch_bn_sums<float, 0>(y, nullptr, work, gm, c, blocks, rpb, det_ws, st);
  }
  int total_rows = gm.grp_rows;
  if (sync && sync->fn && sync->world > 1) {
    const int cnt = gm.groups * 2 * c;
    hipLaunchKernelGGL(bn_sync_pack_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, work, sync->scratch, gm.groups, c, 0);
    sync->fn(sync->user, sync->scratch, (int64_t)cnt, stream);
    hipLaunchKernelGGL(bn_sync_pack_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, work, sync->scratch, gm.groups, c, 1);
    total_rows = gm.grp_rows * sync->world;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, st, work, gamma, beta, running_mean, running_var, gm.groups, total_rows, c, momentum, eps);
  {
    const int per_img = hw * (c / (dtype == VDQN_BF16 ? 8 : 4));
    ProfScope ps("bn_apply", 0.0, elems * esz * (resid ? 3 : 2), st);
    if (dtype == VDQN_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16raw>), dim3(apply_chunks(per_img), n_img), dim3(256), 0, st, (const bf16raw*)y, (const bf16raw*)resid, (bf16raw*)z, work, gm, c, relu);
    else hipLaunchKernelGGL((bn_apply_kernel<float>), dim3(apply_chunks(per_img), n_img), dim3(256), 0, st, (const float*)y, (const float*)resid, (float*)z, work, gm, c, relu);
  }
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_bn_train_fwd(const void* y, const void* resid, void* z, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float* work, int32_t n_img, int32_t hw, int32_t c, int32_t num_frames,
                                 int32_t imgs_per_half, int32_t relu, float momentum, float eps, int32_t dtype, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  BnSync sy = {nullptr, nullptr, nullptr, 1, reinterpret_cast<float*>(workspace), workspace_bytes};
  return vdqn_bn_train_fwd_impl(y, resid, z, gamma, beta, running_mean, running_var, work, n_img, hw, c, num_frames, imgs_per_half, relu, momentum,
                                eps, dtype, stream, workspace ? &sy : nullptr);
}

// dgamma / dbeta receive the LOCAL sums (the flat gradient is summed over the ranks later); dy uses the global ones
int vdqn_bn_train_bwd_impl(const void* g, const void* y, void* dy, float* work, float* dgamma, float* dbeta, int32_t n_img, int32_t hw,
                           int32_t c, int32_t num_frames, int32_t imgs_per_half, int32_t dtype, void* stream, const BnSync* sync) {
  VDQN_CHECK(g && y && dy && work, "vdqn_bn_train_bwd: null arg");
  BnGeom gm;
  if (int rc = make_geom("vdqn_bn_train_bwd", n_img, hw, c, num_frames, imgs_per_half, dtype, &gm)) return rc;
  hipStream_t st = (hipStream_t)stream;
  int rpb;
  const int blocks = grid_rows(gm.grp_rows, &rpb);
  float* det_ws = nullptr;
  if (int rc = det_workspace("vdqn_bn_train_bwd", sync, gm, blocks, c, &det_ws)) return rc;
  const double esz = dtype == VDQN_BF16 ? 2 : 4, elems = (double)n_img * hw * c;
  {
    ProfScope ps("bn_bwd_sums", 0.0, 2.0 * elems * esz, st);
    if (dtype == VDQN_BF16) launch_bn_sums<bf16raw, 1>(g, y, work, gm, c, blocks, rpb, det_ws, st);
    else launch_bn_sums<float, 1>(g, y, work, gm, c, blocks, rpb, det_ws, st);
  }
  if (dgamma || dbeta) hipLaunchKernelGGL(bn_param_grad_kernel, dim3((c + 255) / 256), dim3(256), 0, st, work, dgamma, dbeta, gm.groups, c);
  int total_rows = gm.grp_rows;
  if (sync && sync->fn && sync->world > 1) {
    const int cnt = gm.groups * 2 * c;
    hipLaunchKernelGGL(bn_sync_pack_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, work, sync->scratch, gm.groups, c, 0);
    sync->fn(sync->user, sync->scratch, (int64_t)cnt, stream);
    hipLaunchKernelGGL(bn_sync_pack_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, work, sync->scratch, gm.groups, c, 1);
    total_rows = gm.grp_rows * sync->world;
  }
  {
    const int per_img = hw * (c / (dtype == VDQN_BF16 ? 8 : 4));
    ProfScope ps("bn_bwd_apply", 0.0, 3.0 * elems * esz, st);
    if (dtype == VDQN_BF16) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16raw>), dim3(apply_chunks(per_img), n_img), dim3(256), 0, st, (const bf16raw*)g, (const bf16raw*)y, (bf16raw*)dy, work, gm, c, 1.0f / (float)total_rows);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(apply_chunks(per_img), n_img), dim3(256), 0, st, (const float*)g, (const float*)y, (float*)dy, work, gm, c, 1.0f / (float)total_rows);
  }
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_bn_train_bwd(const void* g, const void* y, void* dy, float* work, float* dgamma, float* dbeta, int32_t n_img, int32_t hw,
                                 int32_t c, int32_t num_frames, int32_t imgs_per_half, int32_t dtype, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  BnSync sy = {nullptr, nullptr, nullptr, 1, reinterpret_cast<float*>(workspace), workspace_bytes};
  return vdqn_bn_train_bwd_impl(g, y, dy, work, dgamma, dbeta, n_img, hw, c, num_frames, imgs_per_half, dtype, stream, workspace ? &sy : nullptr);
}

extern "C" int vdqn_avgpool_fwd(const void* x, void* out, int32_t n_img, int32_t hw, int32_t c, int32_t dtype, void* stream) {
  VDQN_CHECK(x && out && n_img > 0 && hw > 0, "vdqn_avgpool_fwd: bad args");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_avgpool_fwd: bad dtype");
  const int e16 = dtype == VDQN_BF16 ? 8 : 4;
  VDQN_CHECK(c % e16 == 0, "vdqn_avgpool_fwd: channels");
  const int total = n_img * (c / e16);
  ProfScope ps("avgpool_fwd", 0.0, (double)n_img * hw * c * (dtype == VDQN_BF16 ? 2 : 4), (hipStream_t)stream);
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((avgpool_fwd_kernel<bf16raw>), dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)x, (bf16raw*)out, n_img, hw, c);
  else hipLaunchKernelGGL((avgpool_fwd_kernel<float>), dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, n_img, hw, c);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}

extern "C" int vdqn_avgpool_bwd(const void* g, const void* x, void* gx, int32_t n_img, int32_t hw, int32_t c, int32_t dtype, void* stream) {
  VDQN_CHECK(g && x && gx && n_img > 0 && hw > 0, "vdqn_avgpool_bwd: bad args");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_avgpool_bwd: bad dtype");
  const int e16 = dtype == VDQN_BF16 ? 8 : 4;
  VDQN_CHECK(c % e16 == 0, "vdqn_avgpool_bwd: channels");
  const long total = (long)n_img * hw * (c / e16);
  ProfScope ps("avgpool_bwd", 0.0, 2.0 * n_img * hw * c * (dtype == VDQN_BF16 ? 2 : 4), (hipStream_t)stream);
  if (dtype == VDQN_BF16) hipLaunchKernelGGL((avgpool_bwd_kernel<bf16raw>), dim3(grid_elems(total)), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)g, (const bf16raw*)x, (bf16raw*)gx, n_img, hw, c);
  else hipLaunchKernelGGL((avgpool_bwd_kernel<float>), dim3(grid_elems(total)), dim3(256), 0, (hipStream_t)stream, (const float*)g, (const float*)x, (float*)gx, n_img, hw, c);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
