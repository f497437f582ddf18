// Probe: the K-step SCHEDULE of the nine-tap window kernel with its memory side kept (LDS-DMA staging from L2, fragment reads from
// LDS, barriers), real addresses replaced by fixed ones.  Per K-step of a 128 x 128 x 64 tile: 16 KiB of weights + ~2.7 KiB of window
// by LDS-DMA, 16 ds_read_b128 and 32 v_mfma_f32_16x16x32_bf16 per wave.
//   mode 0: the shipped schedule — 4 waves per workgroup, 2 workgroups per CU, two weight buffers, one barrier per K-step,
//           vmcnt(0) + barrier -> issue the staging of step k + 2 -> fragment reads of step k + 1 under the MFMAs of step k
//   mode 1: 8 waves per workgroup, 1 workgroup per CU, a 256 x 128 tile as two 128-row halves that SHARE the weight tile (three-slot
//           ring), the two halves half a K-step apart: one half issues its 32 MFMAs while the other reads its fragments and issues
//           the DMA pieces, a barrier, roles swap ("ping-pong"); counted vmcnt, 1.25-2 K-steps of DMA lead
//   mode 2: mode 1 without the DMA (fragment reads + MFMAs + barriers only)
//   mode 3: mode 0 without the DMA
// Build: hipcc -O3 --offload-arch=gfx950 pingpong.hip -o pingpong ; run: ./pingpong
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kWt = 16384;   // one weight tile: 128 columns x 64 K x bf16
constexpr int kWin = 24576;  // one window buffer (192 rows x 128 B)

__device__ __forceinline__ void dma16(const unsigned char* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

#define MFMA32(A, B)                                                                                                        \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[h][i], B[h][j], acc[i][j], 0, 0, 0);

// ---- mode 0 / 3: the shipped schedule ----
// STRIDED: the weight tile as the shipped kernels read it — 128 rows of 128 B, 4,608 B apart (a [co][9 ci] matrix with ci = 256), the K
// offset moving 128 B per step — instead of 16 KiB contiguous
template <bool DMA, bool STRIDED = false>
__global__ __launch_bounds__(256, 2) void shipped(const unsigned char* __restrict__ wts, unsigned wt_mask, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2 weight tiles][2 windows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (2 * kWt + 2 * kWin) / 4; i += 256) {
    const unsigned hsh = (unsigned)i * 2654435761u;
    ((unsigned*)lds)[i] = 0x3c003c00u ^ (hsh & 0x807f807fu) ^ ((hsh >> 9) & 0x03000300u);  // bf16 values of magnitude 0.25 .. 2, random sign / mantissa
  }
  __syncthreads();
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[2][2][4], b[2][2][4];
  const int i16 = lane & 15, g = lane >> 4;
  const unsigned aoff = (unsigned)(((wave >> 1) * 64 + i16) * 128 + ((g ^ (i16 & 7)) << 4));
  const unsigned boff = (unsigned)(((wave & 1) * 64 + i16) * 128 + ((g ^ (i16 & 7)) << 4));
  auto load = [&](int set, int wbuf, int bbuf, int tap) {
    const unsigned char* wb = lds + 2 * kWt + wbuf * kWin + tap * 128;
    const unsigned char* bt = lds + bbuf * kWt;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      a[set][0][f] = *(const bf16x8*)(wb + f * 2048 + aoff);
      a[set][1][f] = *(const bf16x8*)(wb + f * 2048 + (aoff ^ 64u));
      b[set][0][f] = *(const bf16x8*)(bt + f * 2048 + boff);
      b[set][1][f] = *(const bf16x8*)(bt + f * 2048 + (boff ^ 64u));
    }
  };
  const unsigned lane_g = (unsigned)(tid * 16);
  unsigned char* lw = lds + wave * 1024;
  load(0, 0, 0, 0);
  unsigned gofs = (blockIdx.x & 7u) * 4096u;
  for (int k = 0; k < steps; ++k) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (DMA) {
        if (STRIDED) {
          const int kstep = (2 * k + u) % 36;
          const unsigned char* src = wts + (blockIdx.x & 1u) * (128u * 4608u) + (unsigned)(tid >> 3) * 4608u + (unsigned)kstep * 128u + (unsigned)(tid & 7) * 16u;
#pragma unroll
          for (int pc = 0; pc < 4; ++pc) dma16(src + pc * 32 * 4608, lw + u * kWt + pc * 4096);
          dma16(wts + (2u << 20) + (gofs & wt_mask) + lane_g, lw + 2 * kWt + ((k ^ u) & 1) * kWin + (k & 3) * 4096);
        } else {
          const unsigned char* src = wts + (gofs & wt_mask) + lane_g;
#pragma unroll
          for (int pc = 0; pc < 4; ++pc) dma16(src + pc * 4096, lw + u * kWt + pc * 4096);
          dma16(src + 4 * 4096, lw + 2 * kWt + ((k ^ u) & 1) * kWin + (k & 3) * 4096);  // (a window piece: 1 of ~1.3 per wave and step)
        }
        gofs += kWt;
      }
      __builtin_amdgcn_sched_barrier(0);
      load(u ^ 1, 0, u ^ 1, (k + u) % 3);
      MFMA32(a[u], b[u])
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

// ---- mode 1 / 2: ping-pong ----
template <bool DMA>
__global__ __launch_bounds__(512, 1) void pingpong(const unsigned char* __restrict__ wts, unsigned wt_mask, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [3 weight tiles][2 windows of 320 rows]
  constexpr int kWin2 = 320 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (3 * kWt + 2 * kWin2) / 4; i += 512) ((unsigned*)lds)[i] = 0x3c003c00u;
  __syncthreads();
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[2][4], b[2][4];
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), w4 = wave & 3;
  const int i16 = lane & 15, g = lane >> 4;
  const unsigned aoff = (unsigned)((grp * 128 + (w4 >> 1) * 64 + i16) * 128 + ((g ^ (i16 & 7)) << 4));
  const unsigned boff = (unsigned)(((w4 & 1) * 64 + i16) * 128 + ((g ^ (i16 & 7)) << 4));
  auto load = [&](int slot, int tap) {
    const unsigned char* wb = lds + 3 * kWt + tap * 128;
    const unsigned char* bt = lds + slot * kWt;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      a[0][f] = *(const bf16x8*)(wb + f * 2048 + aoff);
      a[1][f] = *(const bf16x8*)(wb + f * 2048 + (aoff ^ 64u));
      b[0][f] = *(const bf16x8*)(bt + f * 2048 + boff);
      b[1][f] = *(const bf16x8*)(bt + f * 2048 + (boff ^ 64u));
    }
  };
  const unsigned lane_g = (unsigned)((tid & 255) * 16);
  unsigned char* lw = lds + w4 * 1024;
  unsigned gofs = (blockIdx.x & 7u) * 4096u + (unsigned)grp * 8192u;
  int slot = 0;   // ring slot of the step whose fragments this wave reads next
  int islot = 2;  // ring slot this wave's next weight pieces go to
  // memory phase of a wave: the fragments of its next step, then its share of a later step's staging (2 of the 16 weight pieces, 1 window piece)
  auto mem_phase = [&](int k) {
    load(slot, k % 3);
    slot = slot == 2 ? 0 : slot + 1;
    if (DMA) {
      const unsigned char* src = wts + (gofs & wt_mask) + lane_g;
      dma16(src, lw + islot * kWt + grp * 8192);
      dma16(src + 4096, lw + islot * kWt + grp * 8192 + 4096);
      dma16(src + 2 * kWt, lw + 3 * kWt + kWin2 + (k & 7) * 4096 + grp * 2048 * 0);
      gofs += kWt;
      islot = islot == 2 ? 0 : islot + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  if (grp == 0) mem_phase(0);
  __syncthreads();
  for (int k = 0; k < steps; ++k) {
    // half-step 2k: group 0 computes step k, group 1 reads / stages
    if (grp == 0) {
      __builtin_amdgcn_s_setprio(1);
      MFMA32(a, b)
      __builtin_amdgcn_s_setprio(0);
    } else {
      mem_phase(k);
    }
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // half-step 2k + 1: roles swapped
    if (grp == 0) {
      mem_phase(k + 1);
    } else {
      __builtin_amdgcn_s_setprio(1);
      MFMA32(a, b)
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) out[blockIdx.x * 512 + tid] = s;
}

template <typename K>
void run(const char* name, K kern, int threads, size_t smem, int blocks, const unsigned char* wts, unsigned mask, float* out) {
  const int steps = 1500;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, wts, mask, out, 50);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, wts, mask, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // flops: every wave issues 32 MFMAs of 16x16x32 per step (mode 0: two K-steps per loop iteration)
    const double per_wave_step = 32.0 * 2.0 * 16 * 16 * 32;
    const double flops = (double)blocks * (threads / 64) * steps * (threads == 256 ? 2.0 : 1.0) * per_wave_step;
    printf("%-58s blocks %5d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  }
}

int main() {
  float* out;
  unsigned char* wts;
  hipMalloc(&out, 4096 * 512 * 4);
  const size_t wbytes = 8u << 20;
  hipMalloc(&wts, wbytes + (1 << 20));
  {
    unsigned short* h = new unsigned short[(wbytes + (1 << 20)) / 2];
    unsigned x = 12345u;
    for (size_t i = 0; i < (wbytes + (1 << 20)) / 2; ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = (unsigned short)(0x3c00u ^ ((x >> 16) & 0x807fu) ^ ((x >> 8) & 0x0300u));
    }
    hipMemcpy(wts, h, wbytes + (1 << 20), hipMemcpyHostToDevice);
    delete[] h;
  }
  const size_t smem0 = 2 * kWt + 2 * kWin, smem1 = 3 * kWt + 2 * 320 * 128;
  for (unsigned mask : {(1u << 20) - 1, (4u << 20) - 1}) {  // weights of 1 MiB (layer3) / 4 MiB (layer4) streamed over and over
    printf("weight stream wraps at %u MiB\n", (mask + 1) >> 20);
    run("shipped schedule (4 waves x 2 workgroups per CU)", shipped<true>, 256, smem0, 512, wts, mask & ~15u, out);
    run("shipped schedule, weight rows 4608 B apart", shipped<true, true>, 256, smem0, 512, wts, mask & ~15u, out);
    run("shipped schedule, no DMA", shipped<false>, 256, smem0, 512, wts, mask & ~15u, out);
    run("ping-pong (8 waves, shared weight tile, 3-slot ring)", pingpong<true>, 512, smem1, 256, wts, mask & ~15u, out);
    run("ping-pong, no DMA", pingpong<false>, 512, smem1, 256, wts, mask & ~15u, out);
  }
  return 0;
}
