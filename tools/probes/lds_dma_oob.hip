// Probe: does `buffer_load_dwordx4 ... lds` with an out-of-range voffset write zeros into LDS on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const void* in, unsigned* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* s32 = (unsigned*)smem;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s32[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  if ((threadIdx.x % 5) == 0) voff = 0x7fffffff;      // far out of range
  if ((threadIdx.x % 7) == 0) voff = nbytes;          // first byte out of range
  const int wave = threadIdx.x >> 6;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = s32[i];
}
int main() {
  const int n = 4096;
  std::vector<unsigned> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = 0x1000 + i;
  void* d; unsigned* o;
  hipMalloc(&d, n); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, o, n);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0, zeros = 0, good = 0;
  for (int t = 0; t < 256; ++t) {
    bool oob = (t % 5 == 0) || (t % 7 == 0);
    for (int e = 0; e < 4; ++e) {
      unsigned v = r[t * 4 + e];
      if (oob) { if (v == 0) zeros++; else { bad++; if (bad < 5) printf("oob lane %d elem %d = %08x\n", t, e, v); } }
      else { if (v == 0x1000u + t * 4 + e) good++; else { bad++; if (bad < 5) printf("lane %d elem %d = %08x\n", t, e, v); } }
    }
  }
  printf("LDS-DMA OOB probe: good=%d zeros=%d bad=%d\n", good, zeros, bad);
  return bad != 0;
}
