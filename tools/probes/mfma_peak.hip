// Probe: what does the MFMA pipe of this MI355X sustain for the inner-loop SHAPE of the implicit-GEMM kernels, with
// the memory side taken away step by step?  A workgroup = 4 waves, each wave owns a 64x64 accumulator tile (16
// v_mfma_f32_16x16x32_bf16 per K=32 step, 8 operand fragments of 16 B per lane), 2 workgroups per CU, 512 workgroups.
//   mode 0: MFMAs only, operands constant in registers
//   mode 1: + the 8 fragment reads per step from LDS (ds_read_b128, conflict-free rows), register double-buffered
//   mode 2: + one workgroup barrier per two K-steps (one 64-deep staged tile)
//   mode 3: mode 2 with 8 waves per SIMD-pair... (not used)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int steps) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += 256) ((unsigned*)lds)[i] = 0x3c003c00u + (i & 3);
  __syncthreads();
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[2][4], b[2][4];
  // fragment rows: 64-byte rows (K=32 bf16), lane -> row (lane & 15), 16-byte column (lane >> 4); one wave reads 1 KiB rows
  const unsigned char* abase = lds + (wave >> 1) * 8192 + (lane & 15) * 64 + ((lane >> 4) ^ ((lane >> 2) & 3)) * 16;
  const unsigned char* bbase = lds + 32768 + (wave & 1) * 8192 + (lane & 15) * 64 + ((lane >> 4) ^ ((lane >> 2) & 3)) * 16;
  unsigned sel = (unsigned)steps * 2654435761u + lane;  // per-lane junk that the address "selection" VALU chain chews on
  auto load = [&](int buf, int k) {
    unsigned off = 0;
    if (MODE >= 3) {  // ~20 VALU per half step, the cost of the window kernel's tap / edge selection
#pragma unroll
      for (int r = 0; r < 20; ++r) asm("v_mad_u32_u24 %0, %0, 5, %1" : "+v"(sel) : "v"(k + r));
      off = sel & (unsigned)(steps < 0);  // always 0, unknown to the compiler
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[buf][i] = *(const bf16x8*)(abase + off + i * 1024 + (k & 1) * 4096);
      b[buf][i] = *(const bf16x8*)(bbase + off + i * 1024 + (k & 1) * 4096);
    }
  };
  load(0, 0);
  load(1, 1);
  for (int k = 0; k < steps; ++k) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (MODE >= 1) load(half ^ 1, k);
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);  // the shipped kernels: all reads + address VALU first, then the MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[half][i], b[half][j], acc[i][j], 0, 0, 0);
      if (MODE == 4) {  // one DS read + a few VALU behind every second MFMA
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        }
      }
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("" : "+v"(a[half ^ 1][i]));
          asm volatile("" : "+v"(b[half ^ 1][i]));
        }
      }
    }
    if (MODE >= 2) __syncthreads();
  }
  if (MODE >= 3 && sel == 77u) out[0] = 1.f;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, float* out, int blocks) {
  const int steps = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * steps * 2 * 16 * (2.0 * 16 * 16 * 32);
    printf("%-44s blocks %5d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {512, 2048}) {
    run<0>("mfma only", out, blocks);
    run<1>("+ 8 x ds_read_b128 per 16 mfma", out, blocks);
    run<2>("+ barrier per 2 K-steps", out, blocks);
    run<3>("+ 20 VALU/half, reads+VALU BEFORE the mfmas", out, blocks);
    run<4>("same work, interleaved (sched_group_barrier)", out, blocks);
  }
  // sustained: does the clock hold when the MFMA pipe is kept busy for seconds?
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e[41];
    for (auto& x : e) hipEventCreate(&x);
    hipEventRecord(e[0]);
    for (int i = 0; i < 40; ++i) {
      for (int r = 0; r < 15; ++r) {
        if (mode == 0) probe<0><<<2048, 256>>>(out, 2000);
        else probe<2><<<2048, 256>>>(out, 2000);
      }
      hipEventRecord(e[i + 1]);
    }
    hipDeviceSynchronize();
    printf("sustained %s:", mode == 0 ? "mfma only" : "reads + barrier");
    for (int i = 0; i < 40; i += 3) {
      float ms;
      hipEventElapsedTime(&ms, e[i], e[i + 1]);
      printf(" %.0f", 15.0 * 2048 * 4 * 2000 * 2 * 16 * (2.0 * 16 * 16 * 32) / ms / 1e9);
    }
    printf(" TFLOP/s (every third 15-launch window)\n");
  }
  return 0;
}
