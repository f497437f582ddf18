// Paired nine-tap window kernel for the 3x3 / stride 1 / pad 1 convolutions with 128+ channels (ResNet layer2 - layer4, bf16;
// MODE 0 forward, MODE 1 data gradient): the dominant kernel of a TD update.
//
// igemm_win9_kernel (igemm.hip) runs two 4-wave workgroups per CU, each wave sharing its SIMD's matrix pipe with a wave of the
// OTHER workgroup.  K-step stamps (profiles/r02a_win9_kstep_stamps.txt) show what that costs: the two workgroups drift into
// lock-step, so both waves of a SIMD are in their MFMA phase together (32 MFMAs take ~1060 cycles instead of 512) and in
// their staging phase together (~470 cycles with the pipe idle): the loop holds the pipe ~68 % busy.  Two workgroups cannot
// be phase-locked against each other; the two halves of ONE workgroup can (cdna_hip_programming.md, the 256^2 8-phase
// template's staggered wave groups).  Here a workgroup is 8 waves = two GROUPS of 4, each group computing its own 128 x 128
// tile exactly like igemm_win9_kernel (own LDS windows and weight buffers), and every K-step is split into two barrier-
// delimited intervals:
//     N(k): issue the LDS-DMA of step k+1, read the fragments of step k from LDS
//     C(k): 32 MFMAs on those fragments
// Group B enters the loop one barrier late, so while A is in C(k), B is in N(k) and vice versa: a SIMD's matrix pipe always
// belongs to one wave, and the other wave's staging / fragment reads / address arithmetic run underneath.  The fragments of a
// step are complete before its MFMAs start, so there is ONE register set of fragments (64 VGPRs fewer than the double-buffered
// loop).  The epilogue is igemm_epilogue with the thread's index inside its group.
//
// Serves the same reference call sites as igemm.hip: torch conv2d (+ folded BatchNorm, ReLU, residual) of
// archs/HabitatDQNMultiAction.py:30,49-51 (torchvision BasicBlock conv1 / conv2) and their data gradient
// (train_q_network.py:226).
#include <stdlib.h>

#include "igemm_common.h"

namespace {

template <int MODE>
__global__ __launch_bounds__(512, 2) void win9x2_kernel(const IgemmParams p, const int wrows, const int n_pairs_m, const FastDiv d_wo, const FastDiv d_howo,
                                                         void* stamps) {
  static_assert(MODE == 0 || MODE == 1, "window kernel: forward or stride-1 data gradient");
  using T = bf16raw;  // (VDQN_INTERLEAVE keys on sizeof(T))
  constexpr int BM = 128, BN = 128, WN = 2;
  constexpr int NF = BN / (16 * WN);  // 4
  constexpr int CPL = 4 * NF;         // 16
  constexpr int PSTR = 32 * 128;      // LDS distance between a wave's consecutive window pieces (4 waves x 8 rows)
  constexpr int WTILE = BN * 128;     // one staged weight tile: 128 rows x 128 B
  const int WBYTES = wrows * 128;     // wrows: multiple of 32, > 128 + 2 W + 2 (its last row is never sourced: the zero row)
  const int NPW = wrows >> 5;         // window pieces (8 rows x 128 B) per wave of a group: 5 or 6
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int tid_wg = threadIdx.x;
  const int grp = __builtin_amdgcn_readfirstlane(tid_wg >> 8);  // 0 = group A (waves 0-3), 1 = group B (waves 4-7)
  const int tid = tid_wg & 255;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid_wg >> 6);
  // LDS: [group A: 2 windows][group B: 2 windows][weight ring: 4 tiles, shared by both groups]
  unsigned char* sA = smem_all + grp * (2 * WBYTES);
  unsigned char* sW = smem_all + 4 * WBYTES;

  // a workgroup = one column tile x two consecutive row tiles (group A the even one, group B the odd one): both groups
  // multiply by the SAME weight tile, which is staged once for the two of them
  const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lb % (uint32_t)p.tiles_n), pair_m = (int)(lb / (uint32_t)p.tiles_n);
  const int tile_m = 2 * pair_m + grp;
  const bool tile_ok = tile_m < p.tiles_m;  // odd row-tile count: the last workgroup's group B stages weights but owns no rows
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int nk = p.nk;
  const int W = p.wo, H = p.ho, rows_total = tile_ok ? p.M : 0;

  const unsigned long long a_ptr = (unsigned long long)p.in;
  const unsigned long long b_ptr = (unsigned long long)p.wt;
  const i32x4 rs_a = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_ptr), __builtin_amdgcn_readfirstlane((int)((a_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane((int)p.in_bytes), 0x00020000};
  const i32x4 rs_b = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_ptr), __builtin_amdgcn_readfirstlane((int)((b_ptr >> 32) & 0xffff)),
                      __builtin_amdgcn_readfirstlane(p.wt_bytes), 0x00020000};

  // ---- window rows staged by this thread (its group's window): j = lrow + 32 i; rows past 128 + 2 W + 2 and pixels outside the
  // tensor are zero (out-of-range offset) ----
  const int lrow = tid >> 3;
  const int lchunk_a = (tid & 7) ^ (lrow & 7);
  const int pixB = p.pix_stride * 2;
  const int need = BM + 2 * W + 2;
  uint32_t a_off[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int j = lrow + 32 * i;
    const int q = m0 - W - 1 + j;
    a_off[i] = (j < need && (unsigned)q < (unsigned)rows_total) ? (uint32_t)q * (uint32_t)pixB + (uint32_t)(lchunk_a * 16) : kOob;
  }
  // ---- weight rows staged by this thread: wave w of the EIGHT stages rows 16 w .. 16 w + 15 as two 8-row pieces ----
  uint32_t b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 16 * (tid_wg >> 6) + 8 * i + (lane >> 3);
    const int key = (((r / CPL) & 1) << 2) | (r & 3);  // the reader's (i16 & 7): see igemm.hip
    b_off[i] = (uint32_t)(n0 + r) * (uint32_t)(p.ktot * 2) + (uint32_t)(((lane & 7) ^ key) * 16);
  }

  const uint32_t lds_all = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_all;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_win = lds_all + (uint32_t)(grp * (2 * WBYTES)) + (uint32_t)wave_u * (8 * 128);
  const uint32_t lds_wt = lds_all + (uint32_t)(4 * WBYTES) + (uint32_t)wave8 * (16 * 128);

#define VDQN_DMA1(V0, LDS, RSRC, SOFF)                                                                             \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" ::"v"(V0), "s"(LDS), "s"(RSRC), "s"(SOFF) : "memory")
#define VDQN_DMA2(V0, V1, LDS, RSRC, SOFF, STRIDE)                                                                  \
  asm volatile(                                                                                                     \
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %4 offen lds\n\t"                                \
      "s_add_u32 m0, %2, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds"                                 \
      ::"v"(V0), "v"(V1), "s"(LDS), "s"(RSRC), "s"(SOFF), "n"(STRIDE)                                               \
      : "memory", "scc")
  // this group's activation window of channel chunk CC -> window buffer WBUF: NPW (5 or 6) pieces per wave
#define VDQN_ISSUE_AW(WBUF, CC)                                                                                     \
  {                                                                                                                 \
    const uint32_t la_ = lds_win + (uint32_t)(WBUF) * (uint32_t)WBYTES;                                             \
    const int so_a_ = (CC) * 128;                                                                                   \
    VDQN_DMA2(a_off[0], a_off[1], la_, rs_a, so_a_, PSTR);                                                          \
    const uint32_t la2_ = la_ + 2 * PSTR;                                                                           \
    VDQN_DMA2(a_off[2], a_off[3], la2_, rs_a, so_a_, PSTR);                                                         \
    const uint32_t la4_ = la_ + 4 * PSTR;                                                                           \
    VDQN_DMA1(a_off[4], la4_, rs_a, so_a_);                                                                         \
    if (NPW > 5) {                                                                                                  \
      const uint32_t la5_ = la_ + 5 * PSTR;                                                                         \
      VDQN_DMA1(a_off[5], la5_, rs_a, so_a_);                                                                       \
    }                                                                                                               \
  }
  // this wave's two pieces of the weight tile of K-step KSTEP -> ring slot SLOT
#define VDQN_ISSUE_W(SLOT, KSTEP)                                                                                   \
  {                                                                                                                 \
    const uint32_t lw_ = lds_wt + (uint32_t)(SLOT) * WTILE;                                                         \
    const int so_ = (KSTEP)*128;                                                                                    \
    VDQN_DMA2(b_off[0], b_off[1], lw_, rs_b, so_, 8 * 128);                                                         \
  }

  f32x4 acc[4][NF];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave / WN, wc = wave % WN;
  const int i16 = lane & 15, g = lane >> 4;
  // edge bits of this lane's four pixels, 4 bits per fragment f: 1 top row, 2 bottom row, 4 left column, 8 right column
  uint32_t edge16 = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const uint32_t m = (uint32_t)(m0 + wr * 64 + f * 16 + i16);
    const uint32_t rem = m - fastdiv(m, d_howo) * d_howo.div;
    const uint32_t oh = fastdiv(rem, d_wo), ow = rem - oh * d_wo.div;
    const uint32_t e = (oh == 0 ? 1u : 0u) | (oh == (uint32_t)H - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == (uint32_t)W - 1 ? 8u : 0u);
    edge16 |= e << (4 * f);
  }

  const int cpk = p.ci / 64;  // channel chunks; K order (chunk, tap), tap fastest

  u32x4 fa[2][2][4], fb[2][2][NF];  // [register set][K half][fragment]: the set of step k+1 is filled under the MFMAs of step k
  const unsigned char* a_rd = sA + (wr * 64 + i16) * 128;
  const unsigned char* b_rd = sW + (wc * (BN / WN) + (i16 >> 2) * CPL + (i16 & 3)) * 128;
  const int bcoff0 = ((g ^ (i16 & 7)) << 4), bcoff1 = (((g + 4) ^ (i16 & 7)) << 4);
  const int zoff = (wrows - 1) * 128 + (g << 4);  // the zero row of a window buffer
  // fragments of K-step (CC, TAP) from weight slot SLOT: the forward reads input pixel m + (kr-1) W + (ks-1), the data gradient
  // m + (1-kr) W + (1-ks)
#define VDQN_LOAD_FRAGS(SET, SLOT, CC, TAP)                                                                              \
  {                                                                                                                      \
    const int kr_ = ((TAP)*11) >> 5, ks_ = (TAP)-3 * kr_; /* TAP / 3 for 0..8 */                                         \
    const int ky_ = MODE == 0 ? kr_ : 2 - kr_, kx_ = MODE == 0 ? ks_ : 2 - ks_;                                          \
    const uint32_t tb_ = (ky_ == 0 ? 1u : 0u) | (ky_ == 2 ? 2u : 0u) | (kx_ == 0 ? 4u : 0u) | (kx_ == 2 ? 8u : 0u);       \
    const uint32_t zm_ = edge16 & (tb_ * 0x1111u);                                                                       \
    const int off_ = W * ky_ + kx_;                                                                                      \
    const int key_ = (i16 + off_) & 7;                                                                                   \
    const int ac0_ = ((g ^ key_) << 4), ac1_ = (((g + 4) ^ key_) << 4);                                                  \
    const unsigned char* w_ = sA + ((CC)&1) * WBYTES;                                                                    \
    const unsigned char* a_ = a_rd + ((CC)&1) * WBYTES + off_ * 128;                                                     \
    const unsigned char* b_ = b_rd + (SLOT)*WTILE;                                                                       \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                                                   \
      const bool z_ = ((zm_ >> (4 * f_)) & 15u) != 0u;                                                                   \
      fa[SET][0][f_] = *reinterpret_cast<const u32x4*>(z_ ? w_ + zoff : a_ + f_ * 16 * 128 + ac0_);                      \
      fa[SET][1][f_] = *reinterpret_cast<const u32x4*>(z_ ? w_ + zoff + 64 : a_ + f_ * 16 * 128 + ac1_);                 \
    }                                                                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                  \
      fb[SET][0][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + bcoff0);                                      \
      fb[SET][1][j_] = *reinterpret_cast<const u32x4*>(b_ + j_ * 4 * 128 + bcoff1);                                      \
    }                                                                                                                    \
  }
#define VDQN_MFMA_ALL(SET)                                                                                               \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)                     \
      _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) {                                                                \
    acc[f_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[SET][h_][j_]),                   \
                                                          __builtin_bit_cast(bf16x8, fa[SET][h_][f_]), acc[f_][j_], 0, 0, 0); \
  }
#ifdef VDQN_STAMP
  unsigned long long st_n = 0, st_nwait = 0, st_nbar = 0, st_c = 0, st_cbar = 0, st_t = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long st_rt_begin = __builtin_amdgcn_s_memrealtime();
#define VDQN_ST(ACC)                                                     \
  {                                                                      \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();          \
    ACC += n_ - st_t;                                                    \
    st_t = n_;                                                           \
  }
#else
#define VDQN_ST(ACC)
#endif

  // prologue: weight tiles of K-steps 0, 1, 2 (this wave's pieces), this group's window of chunk 0; visible to everyone;
  // fragments of step 0
  VDQN_ISSUE_W(0, 0)
  VDQN_ISSUE_W(1, cpk)     /* step 1 = (chunk 0, tap 1): weight K offset tap * cpk + chunk */
  VDQN_ISSUE_W(2, 2 * cpk)
  VDQN_ISSUE_AW(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  VDQN_LOAD_FRAGS(0, 0, 0, 0)
  if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: group B runs one interval behind group A
#ifdef VDQN_STAMP
  st_t = __builtin_amdgcn_s_memtime();
#endif
  int cc = 0, tap = 0;                  // K-step k = (cc, tap); its weight tile sits in ring slot k % 4
  int l_cc = 0, l_tap = 1, l_slot = 1;  // K-step k + 1: its fragments are read during C(k)
  int i_cc = 0, i_tap = 3, i_slot = 3;  // K-step k + 3: its weight tile is staged during N(k)
  // one K-step = two barrier-delimited intervals:
  //   N(k): stage this wave's two pieces of the weight tile of step k+3 into the ring slot step k-1 used (its last readers, the
  //         two groups' C(k-2), are two and three intervals back) and, at tap 1, this group's window of the next chunk; then wait
  //         for everything staged EARLIER (counted: the pieces issued just now stay in flight for two more intervals)
  //   C(k): the MFMAs of step k, the matrix pipe being this group's alone; the fragment reads of step k+1 issue in their shadow
#define VDQN_STEP(K, CUR, NXT)                                                                                           \
  {                                                                                                                      \
    const bool more_w = (K) + 3 < nk;                                                                                    \
    const bool more_a = tap == 1 && cc + 1 < cpk;                                                                        \
    if (more_w) VDQN_ISSUE_W(i_slot, i_tap * cpk + i_cc)                                                                 \
    if (more_a) VDQN_ISSUE_AW((cc + 1) & 1, cc + 1)                                                                      \
    VDQN_ST(st_n)                                                                                                        \
    if (!more_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
    else if (!more_a) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                   \
    else if (NPW == 5) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                                  \
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                                \
    VDQN_ST(st_nwait)                                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                        \
    VDQN_ST(st_nbar)                                                                                                     \
    asm volatile("" : "+v"(fa[CUR][0][0]), "+v"(fa[CUR][0][1]), "+v"(fa[CUR][0][2]), "+v"(fa[CUR][0][3]),                \
                      "+v"(fa[CUR][1][0]), "+v"(fa[CUR][1][1]), "+v"(fa[CUR][1][2]), "+v"(fa[CUR][1][3]));               \
    _Pragma("unroll") for (int j_ = 0; j_ < NF; ++j_) asm volatile("" : "+v"(fb[CUR][0][j_]), "+v"(fb[CUR][1][j_]));     \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    VDQN_LOAD_FRAGS(NXT, l_slot, l_cc, l_tap) /* unconditional (the step behind the last re-reads live buffers): one block with the MFMAs */ \
    VDQN_MFMA_ALL(CUR)                                                                                                   \
    VDQN_INTERLEAVE(8 + 2 * NF)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of step k+1's buffers are complete */         \
    VDQN_ST(st_c)                                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                                        \
    VDQN_ST(st_cbar)                                                                                                     \
    if (++tap == 9) {                                                                                                    \
      tap = 0;                                                                                                           \
      ++cc;                                                                                                              \
    }                                                                                                                    \
    if (++l_tap == 9) {                                                                                                  \
      l_tap = 0;                                                                                                         \
      ++l_cc;                                                                                                            \
    }                                                                                                                    \
    if (++i_tap == 9) {                                                                                                  \
      i_tap = 0;                                                                                                         \
      ++i_cc;                                                                                                            \
    }                                                                                                                    \
    l_slot = (l_slot + 1) & 3;                                                                                           \
    i_slot = (i_slot + 1) & 3;                                                                                           \
  }
  for (int k = 0; k < nk; k += 2) {  // nk = 9 * (ci / 64) with ci a multiple of 128: even
    VDQN_STEP(k, 0, 1)
    VDQN_STEP(k + 1, 1, 0)
  }
#undef VDQN_STEP
  if (grp == 0) __builtin_amdgcn_s_barrier();  // pairs with group B's last interval
#undef VDQN_LOAD_FRAGS
#undef VDQN_MFMA_ALL
#undef VDQN_ISSUE_AW
#undef VDQN_ISSUE_W
#undef VDQN_DMA2
#undef VDQN_DMA1
#ifdef VDQN_STAMP
  const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
  // the epilogue's LDS scratch (column sums) is this group's first window buffer: nobody reads a window any more
  igemm_epilogue<T, BM, BN, MODE, WN>(p, acc, sA, m0, tile_ok ? n0 : p.co, tile_m, rows_total, p.howo, W, 0, 0, tid);
#ifdef VDQN_STAMP
  if (stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = reinterpret_cast<unsigned long long*>(stamps) + ((size_t)blockIdx.x * 2 + grp) * 16;
    o[0] = st_begin; o[1] = st_loop_end; o[2] = __builtin_amdgcn_s_memtime();
    o[3] = st_n; o[4] = st_nbar; o[5] = st_c; o[6] = st_nwait; o[7] = (unsigned long long)nk;
    o[8] = st_rt_begin; o[9] = __builtin_amdgcn_s_memrealtime();
    o[10] = st_cbar;
  }
#endif
#undef VDQN_ST
}

}  // namespace

#ifdef VDQN_STAMP
extern void* g_stamp_buffer;
#endif

// entry used by vdqn_conv2d (igemm.hip): returns VDQN_OK or an error code
int vdqn_launch_win9x2(const void* pv, int mode, hipStream_t stream) {
  const IgemmParams& p = *reinterpret_cast<const IgemmParams*>(pv);
  const int wrows = (128 + 2 * p.wo + 2 + 1 + 31) & ~31;
  const size_t smem = (size_t)4 * wrows * 128 + 4 * 128 * 128;
  const int n_pairs_m = (p.tiles_m + 1) / 2;
  const unsigned grid = (unsigned)(n_pairs_m * p.tiles_n);
  void* stamps = nullptr;
#ifdef VDQN_STAMP
  stamps = g_stamp_buffer;
#endif
  vdqn_prof_begin(mode == 0 ? "igemm_win<bf16,128,fwd>" : "igemm_win<bf16,128,dgrad>", 2.0 * p.M * p.co * p.ktot,
                  2.0 * ((double)p.n_img * p.hi * p.wi * p.ci + (double)p.co * p.ktot + (double)p.M * p.co * (1 + (p.resid != nullptr) + (p.mask != nullptr))), stream);
  if (mode == 0) {
    vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9x2_kernel<0>), smem);
    hipLaunchKernelGGL((win9x2_kernel<0>), dim3(grid), dim3(512), smem, stream, p, wrows, n_pairs_m, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo), stamps);
  } else {
    vdqn_ensure_dyn_smem(reinterpret_cast<const void*>(&win9x2_kernel<1>), smem);
    hipLaunchKernelGGL((win9x2_kernel<1>), dim3(grid), dim3(512), smem, stream, p, wrows, n_pairs_m, make_fastdiv((uint32_t)p.wo), make_fastdiv((uint32_t)p.howo), stamps);
  }
  vdqn_prof_end(stream);
  VDQN_LAUNCH_CHECK();
  return VDQN_OK;
}
