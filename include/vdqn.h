/*
 * vdqn.h — C ABI of libvdqn.so, the MI355X (gfx950) implementation of the Q-learning hot path of
 * uiuc-robovision/video-dqn (train_q_network.py's inner loop).
 *
 * The reference is pure Python on PyTorch: it has no FFI of its own.  Every entry point below names the
 * reference call site (file:line under the reference tree) whose arithmetic it replaces; the reference-side
 * binding a maintainer would add is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless a name ends in _host;
 *   - the caller owns every buffer (the library allocates nothing on the device except inside an explicit
 *     vdqn_net_create/vdqn_net_destroy pair, which holds host-side descriptors only);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no host sync;
 *   - return value 0 = ok, negative = error (vdqn_last_error() gives the text); no exceptions cross the ABI;
 *   - activations are NHWC; `dtype` selects the storage/compute type of activations and packed weights:
 *       VDQN_F32  : f32 storage, exact-f32 MFMA (v_mfma_f32_16x16x4_f32)   — the parity mode
 *       VDQN_BF16 : bf16 storage, v_mfma_f32_16x16x32_bf16, f32 accumulate — the throughput mode
 *     master parameters, gradients, Adam state and Q-values are always f32.
 */
#ifndef VDQN_H_
#define VDQN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDQN_F32 0
#define VDQN_BF16 1

#define VDQN_OK 0
#define VDQN_ERR_INVALID (-1)
#define VDQN_ERR_LAUNCH (-2)

/* thread-local text of the last error returned on this host thread */
const char* vdqn_last_error(void);
/* ABI version; bumped on any signature change */
int vdqn_abi_version(void);
/* sizeof() of the argument structs as THIS library was compiled (which: 0 vdqn_conv_args, 1 vdqn_wgrad_args, 2 vdqn_td_args,
 * 3 vdqn_net_config, 4 vdqn_param_info, 5 vdqn_prof_entry, 6 vdqn_step_args; -1 for any other value): a foreign-function binding
 * (ctypes / cgo / JNI) compares them with its own struct definitions at load time instead of finding a drifted field at run time */
int32_t vdqn_abi_struct_size(int32_t which);

/* Launch profiler (off by default).  While enabled, every kernel launch of the library is bracketed by HIP
 * events on its launch stream; collect() synchronises those events and returns one entry per kernel symbol:
 * launches, summed device milliseconds, summed algorithmic FLOPs and algorithmic bytes.  Resets on collect. */
typedef struct vdqn_prof_entry {
  char name[48];
  int64_t launches;
  double ms;
  double flops;
  double bytes;
} vdqn_prof_entry;
int vdqn_profile_enable(int on);
int vdqn_profile_collect(vdqn_prof_entry* out, int max_entries);

/* ------------------------------------------------------------------------------------------------
 * Operator level (each is also what the engine below launches).
 * ------------------------------------------------------------------------------------------------ */

/* Implicit-GEMM convolution / linear layer, forward or data-gradient.
 *   out[m, n] = epilogue( sum_{r,s,c} in[pix(m,r,s), c] * wt[n][r][s][c] )
 *   mode 0 (forward gather):  hi = ho*stride - pad + r,           wi likewise
 *   mode 1 (dgrad gather):    hi = (ho + pad - r) / stride if divisible and in range (transposed conv)
 *   epilogue: v += bias[n]; v += resid[m,n]; relu; v = mask[m,n] > 0 ? v : 0   (each optional);
 *             optional per-tile column sums of the stored values (colsum_part)
 * Replaces: torch conv2d/linear (+ folded eval BatchNorm, ReLU, residual add) reached from
 * archs/HabitatDQNMultiAction.py:30-31,49-53 and their autograd backward (train_q_network.py:226).
 * `ci` must be a multiple of 128 bytes / sizeof(elem); `wt` holds co_pad = roundup(co, 64 or 128) rows. */
typedef struct vdqn_conv_args {
  const void* in;      /* [n_img][hi][wi] pixels of pix_stride elems */
  const void* wt;      /* [co_pad][r][s][ci], K-contiguous */
  const float* bias;   /* [co_pad] or NULL */
  const void* resid;   /* [m][ldo] or NULL */
  const void* mask;    /* [m][ldo] or NULL */
  void* out;           /* [m][ldo], dtype; may be NULL if out_f32 is given */
  float* out_f32;      /* optional f32 copy of the result [m][ldo] */
  float* colsum_part;  /* optional [T][ldo] f32, T = ceil(m/R) with R = vdqn_conv2d_colsum_rows(a) = 128 for the tiled kernels
                          (stride-2 dgrad: 4*ceil(m/4/128), one run of tiles per output parity class) and 32 for the Q-head's
                          skinny GEMMs: per R-row tile, column sums of the stored `out` values
                          (summed over tiles they are the bias / BatchNorm-shift gradient of the layer that `out`
                          is the output-gradient of) */
  int32_t n_img, hi, wi, ci, pix_stride;
  int32_t ho, wo, co, ldo;
  int32_t r, s, stride, pad;
  int32_t mode, relu, dtype;
  /* Optional sibling 1x1 / stride-2 / pad-0 convolution fused into a 3x3 / stride-2 / pad-1 call: the downsample branch of a
   * ResNet BasicBlock (torchvision resnet.py `downsample`; archs/HabitatDQNMultiAction.py:30), whose input pixel is the 3x3's
   * centre tap.  All NULL / 0 = no sibling.
   *   mode 0: out2[m, :co2] = relu2?( sum_c in[pix(m, centre), c] * wt2[n][c] + bias2[n] ), wt2 = [co2_pad][1][1][ci];
   *           one launch, the input rows are fetched once for both convolutions.
   *   mode 1: out += dgrad_1x1_stride2(in2, wt2) before the epilogue (residual / mask / column sums then see the sum):
   *           in2 = gradient of the sibling's output, [n_img][hi][wi] pixels of pix_stride elems holding ci2 channels,
   *           wt2 = [co_pad][1][1][ci2] (the sibling's data-gradient operand); bias2 / out2 / relu2 unused. */
  const void* in2;
  const void* wt2;
  const float* bias2;
  void* out2;
  int32_t co2, ldo2, relu2, ci2;
} vdqn_conv_args;
int vdqn_conv2d(const vdqn_conv_args* a, void* stream);
/* rows of `out` covered by one entry of colsum_part for this call (what sizes that buffer): 128, or the row tile of the skinny
 * GEMM kernels that take the Q-head's bf16 linear layers (archs/HabitatDQNMultiAction.py:31; csrc/skinny.hip) */
int32_t vdqn_conv2d_colsum_rows(const vdqn_conv_args* a);

/* Weight gradient of the same layer:  dw[n][r][s][c] += sum_m gy[m, n] * x[pix(m,r,s), c]
 * (split over `splitk` pixel ranges whose partial tiles are added into dw with f32 atomics).  If dbias != NULL it also
 * accumulates dbias[n] += sum_m gy[m, n].
 * Deterministic mode (the reference pins cudnn.deterministic = True, train_q_network.py:88-89): with `workspace` set
 * (vdqn_conv2d_wgrad_workspace_bytes gives the size) every split stores its partial tile into its own copy of dw with plain
 * stores and a second kernel adds the copies to dw in split order — the result no longer depends on the order in which the
 * blocks finish, two runs are bit-identical.
 * Replaces: the convolution_backward/addmm weight-gradient kernels behind loss.backward(),
 * train_q_network.py:226. */
typedef struct vdqn_wgrad_args {
  const void* gy;   /* [m][ldg] gradient w.r.t. the layer output (already ReLU-masked) */
  const void* x;    /* layer input, as vdqn_conv_args.in */
  float* dw;        /* [co_pad][r][s][ci] f32, pre-zeroed */
  float* dbias;     /* [co_pad] f32, pre-zeroed, or NULL */
  int32_t n_img, hi, wi, ci, pix_stride;
  int32_t ho, wo, co, ldg;
  int32_t r, s, stride, pad;
  int32_t splitk, dtype;
  void* workspace;          /* NULL: atomics; else >= vdqn_conv2d_wgrad_workspace_bytes(a) bytes, 16-byte aligned */
  int64_t workspace_bytes;
} vdqn_wgrad_args;
int vdqn_conv2d_wgrad(const vdqn_wgrad_args* a, void* stream);
int64_t vdqn_conv2d_wgrad_workspace_bytes(const vdqn_wgrad_args* a);  /* -1 on invalid arguments */

/* Host side of the streaming input path (decoded-frame shards that do not fit in HBM): gather n records of bytes_each bytes —
 * record i from src[i], e.g. a frame inside a memory-mapped shard — into dst + i * bytes_each (a pinned staging buffer) on
 * `threads` host threads; returns when all are in place.  One copy per frame where the reference's loader makes three
 * (worker decode + stack, collate, pin: dataloaders/q_learning_real.py:55-73 under torch's DataLoader, train_q_network.py:98,114).
 * No GPU involved. */
int vdqn_host_gather(void* dst, const void* const* src, int64_t n, int64_t bytes_each, int32_t threads);

/* Input packing: normalise + space-to-depth the 224x224 RGB frames into the conv1 operand
 * [n][115][115][16] (2x2x3 -> 12 channels + 4 zero, 2-pixel zero border top/left, 1 bottom/right), so the
 * 7x7/2 stem is a 4x4/1 implicit GEMM with 128-byte contiguous K rows.
 *   src_kind 0: uint8 NHWC [n][224][224][3]  -> (x/255 - mean)/std   (util/torch.py:5-12, :26-36)
 *   src_kind 1: f32 NCHW [n][3][224][224] already normalised (the tensor process_batch moves to the
 *               device, train_q_network.py:127-129) */
int vdqn_pack_input(const void* src, int32_t src_kind, void* dst, int32_t n_img, int32_t dtype, void* stream);

/* 3x3/2 pad 1 max-pool, NHWC (torchvision resnet stem; archs/HabitatDQNMultiAction.py:30). idx[m][c] stores
 * the arg-max tap (kh*3+kw, first maximum wins as in torch). */
int vdqn_maxpool_fwd(const void* in, void* out, uint8_t* idx, int32_t n_img, int32_t hi, int32_t wi, int32_t c,
                     int32_t dtype, void* stream);
/* gx[n,h,w,c] = (x[n,h,w,c] > 0) * sum over windows whose arg-max is (h,w) of gy  (pool + ReLU backward).
 * x may be NULL when gy is already zero wherever the pooled value is <= 0 (the engine's case: the producing dgrad
 * masks with the pooled activation; the arg-max of a window holds exactly that value), which saves the read of x. */
int vdqn_maxpool_bwd(const void* gy, const uint8_t* idx, const void* x, void* gx, int32_t n_img, int32_t hi,
                     int32_t wi, int32_t c, int32_t dtype, void* stream);

/* Fused Double-DQN target + TD loss + dLoss/dQ  (train_q_network.py:134-169,180).
 *   Qb = q_before[b,c,act[b]]; a* = argmax_a q_after_online[b,c,:] (first max); Qa = q_after_target[b,c,a*]
 *   Qa *= (1 - term); y = linear ? rew + (Qa - 0.1) : rew + gamma*Qa; rect clip -> clamp(y,0,1)
 *   d = Qb - y;  loss_kind 0 (the reference, :167): l = 0.5 d^2, dl = d
 *                loss_kind 1 (Huber / smooth-L1 with beta 1, the option archs/HabitatDQNMultiAction.py:25 leaves as a
 *                TODO): l = |d| < 1 ? 0.5 d^2 : |d| - 0.5, dl = clamp(d, -1, 1)
 *   l *= valid if use_valid; loss += inv_count * sum l;
 *   dq[b, c*A+a] = (a == act[b]) ? dl * (valid) * inv_count : 0   (rows of ldq elems, zero padded)
 * q_* are f32 [batch][ldq]; act i64; rew/term/valid f32 [batch][n_cat]; loss is a pre-zeroed f32 scalar. */
typedef struct vdqn_td_args {
  const float* q_before;
  const float* q_after_online;
  const float* q_after_target;
  const int64_t* act;
  const float* rew;
  const float* term;
  const float* valid;
  float* loss;
  void* dq;         /* [batch][ldq] dtype */
  float* dq_f32;    /* optional [batch][ldq] */
  int32_t batch, n_cat, n_act, ldq;
  float gamma, inv_count;
  int32_t clip_rect, linear, use_valid, dtype;
  int32_t loss_kind;      /* 0 = half squared error (reference), 1 = Huber */
  int32_t deterministic;  /* 1: the loss is summed by ONE block in a fixed order (no cross-block atomics) */
  float* q_copy;          /* optional [batch][n_cat * n_act] f32: a compact copy of q_before's first n_cat * n_act columns (what the
                             training loop reads back as Q(s)), written by the same launch */
} vdqn_td_args;
int vdqn_td_loss(const vdqn_td_args* a, void* stream);

/* Ground-truth branch (train_q_network.py:170-178): l = 0.5 (Qb*mask - gt)^2, mask = !isnan(gt) when
 * value_learning, else l = 0.5 (Qb - gt)^2.  gt is f32 [batch][n_cat] (NaN allowed). */
int vdqn_gt_loss(const float* q_before, const int64_t* act, const float* gt, float* loss, void* dq, float* dq_f32,
                 int32_t batch, int32_t n_cat, int32_t n_act, int32_t ldq, float inv_count, int32_t value_learning,
                 int32_t dtype, void* stream);

/* The ResNet stem in one kernel: conv1 7x7/2 (as the 4x4/1 convolution over the packed operand of vdqn_pack_input,
 * weights [64][4][1][64] with BatchNorm folded, f32 bias[64]) + ReLU + MaxPool2d(3, 2, 1)
 * (torchvision resnet.py conv1/bn1/relu/maxpool; archs/HabitatDQNMultiAction.py:30).  Writes pool [n][56][56][64] and
 * the argmax codes idx (as vdqn_maxpool_fwd); the 112x112x64 convolution output is never stored.  Results are
 * bit-identical to vdqn_conv2d followed by vdqn_maxpool_fwd.  idx may be NULL for frames that never see a backward pass
 * (the target pass, whose result is `.detach()`ed, and the s' rows of the online pass, which only feed an argmax: train_q_network.py:140-142,148,155-156 — the reference builds their graphs and never back-propagates through them): pool is the same,
 * the arg-max bytes are not computed. */
int vdqn_stem_conv_pool(const void* t_in, const void* wt, const float* bias, void* pool, void* idx, int32_t n_img,
                        int32_t dtype, void* stream);
/* The same with arg-max bytes for the first n_idx_img images only (0 <= n_idx_img <= n_img; idx may be NULL when it is 0): one
 * launch over [s; s'] of the online pass, where only the s rows see loss.backward() (train_q_network.py:131,142,226). */
int vdqn_stem_conv_pool_n(const void* t_in, const void* wt, const float* bias, void* pool, void* idx, int32_t n_img,
                          int32_t n_idx_img, int32_t dtype, void* stream);

/* conv1's weight gradient straight from the POOLED gradient (the extra_capacity stem in bf16): what vdqn_maxpool_bwd(g_pool, idx)
 * followed by vdqn_conv2d_wgrad on the packed stem geometry computes, without the 112 x 112 x 64 gradient of conv1's output
 * ever being stored (autograd's max_pool2d backward + conv1 weight gradient behind loss.backward(), train_q_network.py:226;
 * torchvision resnet stem reached from archs/HabitatDQNMultiAction.py:30).  g_pool [n][56][56][64] bf16, idx as written by
 * vdqn_stem_conv_pool / vdqn_maxpool_fwd, t_in the packed frames of vdqn_pack_input; dw [64][4][64] f32 is ACCUMULATED into
 * (f32 atomics; with a workspace of vdqn_stem_wgrad_pool_workspace_bytes(n_img) bytes: per-block partial copies summed in
 * block order = deterministic).  The gradient tiles the kernel builds in LDS are bit-identical to vdqn_maxpool_bwd's output. */
int vdqn_stem_wgrad_pool(const void* g_pool, const uint8_t* idx, const void* t_in, float* dw, int32_t n_img, void* workspace,
                         int64_t workspace_bytes, void* stream);
int64_t vdqn_stem_wgrad_pool_workspace_bytes(int32_t n_img);

/* Row-wise softmax over the first n_valid columns of x[rows][ld] (f32), written to y[rows][ld] (other columns 0):
 * `torch.softmax(x, dim=1)` of the inverse-action model's 3-way output (archs/inverse_action2.py:95). n_valid <= 64. */
int vdqn_softmax_rows(const float* x, float* y, int32_t rows, int32_t ld, int32_t n_valid, void* stream);

/* Training of the inverse-action model (train_inverse_model.py:86-112): mean nn.CrossEntropyLoss over `rows` samples of
 * f32 logits[rows][ld] (first n_cls columns) and its gradient (softmax - onehot) * inv_count as `dtype` elements;
 * dropout with a caller-supplied 0/1 mask: out = x * mask * scale (forward and backward); y += alpha * x (Adam's
 * weight_decay term, train_inverse_model.py:190). */
int vdqn_softmax_ce(const float* logits, const int64_t* labels, float* loss, void* dlogits, int32_t rows, int32_t ld,
                    int32_t n_cls, float inv_count, int32_t dtype, void* stream);
int vdqn_mask_scale(const void* x, const void* mask, void* out, int64_t n, float scale, int32_t dtype, void* stream);
int vdqn_axpy(float* y, const float* x, float alpha, int64_t n, void* stream);

/* torch.optim.Adam step (train_q_network.py:124,227) over one flat f32 range:
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * Hyper-parameters are doubles like torch's python scalars (1-b2 is formed in double before rounding to f32). */
int vdqn_adam(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, double beta1,
              double beta2, double eps, void* stream);

/* torch.nn.BatchNorm2d in train mode over an NHWC conv output y[n_img][hw][c] (ARCHITECTURE='basic':
 * archs/HabitatDQNMultiAction.py:32-34,37-40 keeps the ResNet in train mode).  Images are sample-major, frame-minor;
 * image i belongs to statistic group (i / imgs_per_half) * num_frames + i % num_frames — one group per model call and
 * frame slot, the minibatches the reference normalises over (:49-51; train_q_network.py:131,142).
 *   z = relu?( (y - mean_g) * rstd_g * gamma + beta (+ resid) );   biased variance, eps inside the sqrt
 *   running_* (may be NULL): updated once per group, in group order, with `momentum` and the unbiased variance
 * work: f32 [groups][6][c] = mean, rstd, scale, shift and two sums; the forward fills it, the backward reads it. */
int vdqn_bn_train_fwd(const void* y, const void* resid, void* z, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float* work, int32_t n_img, int32_t hw, int32_t c,
                      int32_t num_frames, int32_t imgs_per_half, int32_t relu, float momentum, float eps, int32_t dtype,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* dy = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) per group; dgamma/dbeta (may be NULL) = sums over all
 * groups of g*xhat / g.  dy may alias g.  `work` must be the array the forward of the same tensor filled. */
int vdqn_bn_train_bwd(const void* g, const void* y, void* dy, float* work, float* dgamma, float* dbeta, int32_t n_img,
                      int32_t hw, int32_t c, int32_t num_frames, int32_t imgs_per_half, int32_t dtype, void* workspace,
                      int64_t workspace_bytes, void* stream);
/* Deterministic mode of the two calls above (train_q_network.py:88-89 pins cudnn.deterministic): with `workspace` (>= this many
 * bytes, 16-byte aligned; NULL = f32 atomics) every block stores its partial statistic sums and a second kernel adds them in
 * block order — two runs are bit-identical.  -1 on invalid sizes. */
int64_t vdqn_bn_train_workspace_bytes(int32_t n_img, int32_t hw, int32_t c, int32_t num_frames, int32_t imgs_per_half);
/* AdaptiveAvgPool2d(1) of the ResNet (torchvision resnet.py avgpool) on NHWC x[n_img][hw][c] -> out[n_img][c], and its
 * backward fused with the mask of the ReLU that produced x: gx = (x > 0) * g / hw. */
int vdqn_avgpool_fwd(const void* x, void* out, int32_t n_img, int32_t hw, int32_t c, int32_t dtype, void* stream);
int vdqn_avgpool_bwd(const void* g, const void* x, void* gx, int32_t n_img, int32_t hw, int32_t c, int32_t dtype,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Engine level: the whole HabitatDQNMultiAction network and one TD update.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vdqn_net_config {
  int32_t action_dim;      /* A: 3, or 1 for VALUE_LEARNING/ONE_ACTION (train_q_network.py:38-41) */
  int32_t num_classes;     /* 5 */
  int32_t num_frames;      /* F: 1, or 4 for PANORAMA/PREVIOUS_IMAGES (archs/...:16-19); any F >= 1 accepted */
  int32_t extra_capacity;  /* 1: ARCHITECTURE 'extra_capacity' (BatchNorm on running stats, conv+MLP head);
                              0: 'basic' (defaults.py:14 — train-mode BatchNorm, average pool + one Linear) */
  int32_t dtype;           /* VDQN_F32 | VDQN_BF16 */
  int32_t max_batch;       /* largest per-call sample count B the workspaces are sized for */
  int32_t deterministic;   /* 1: run-to-run bit-identical updates (the reference's cudnn.deterministic = True,
                              train_q_network.py:88-89): weight gradients through the two-stage ordered reduction of
                              vdqn_conv2d_wgrad (its workspace is part of `bwd`), the loss summed by one block,
                              ARCHITECTURE='basic': the train-mode BatchNorm statistics through the ordered two-stage sums
                              of vdqn_bn_train_fwd/bwd (workspace inside `acts`). */
} vdqn_net_config;

typedef struct vdqn_net vdqn_net;

int vdqn_net_create(const vdqn_net_config* cfg, vdqn_net** out);
/* The engine runs weight gradients and the target-network forward on a second, internal HIP stream (joined back
 * into the caller's stream before their results are consumed).  on = 0 serialises everything on the caller's
 * stream (used for per-kernel roofline timing, where each kernel must have the chip to itself).  Default: on
 * (or off when the environment has VDQN_NO_OVERLAP=1). */
int vdqn_net_set_overlap(vdqn_net* net, int on);
void vdqn_net_destroy(vdqn_net* net);

/* Parameter table: the reference's named_parameters()/buffers as slices of two flat f32 arrays.
 * kind: 0 trainable parameter (offset into `params`, gradient/Adam state at the same offset),
 *       1 frozen parameter (resnet.fc.*: never receives a gradient; stored after the trainable range),
 *       2 BatchNorm running_mean, 3 running_var (offsets into `bnstats`).
 * `param_id` is the index in the reference's model.parameters() order (Adam state_dict ids). */
typedef struct vdqn_param_info {
  char name[96];
  int64_t offset;
  int64_t numel;
  int32_t ndim;
  int32_t shape[4];
  int32_t kind;
  int32_t param_id;
  int32_t stage;     /* backward stage (0 head+layer4, 1 layer3, 2 layer2+layer1+stem) that completes its grad */
} vdqn_param_info;
int vdqn_net_num_params(const vdqn_net* net);
int vdqn_net_param_info(const vdqn_net* net, int index, vdqn_param_info* out);
int64_t vdqn_net_params_numel(const vdqn_net* net);     /* total flat params (trainable + frozen) */
int64_t vdqn_net_trainable_numel(const vdqn_net* net);  /* prefix that has gradients / Adam state */
int64_t vdqn_net_bnstats_numel(const vdqn_net* net);
/* [begin,end) of the flat gradient completed by backward stage s (for bucketed all-reduce) */
int vdqn_net_stage_range(const vdqn_net* net, int stage, int64_t* begin, int64_t* end);

/* Workspace sizes in bytes.  packed = folded/packed weights of ONE network instance (online or target);
 * acts = activations of one forward over `n_samples` samples; bwd = gradient workspaces of one backward. */
int64_t vdqn_net_packed_bytes(const vdqn_net* net);
int64_t vdqn_net_acts_bytes(const vdqn_net* net, int32_t n_samples);
int64_t vdqn_net_bwd_bytes(const vdqn_net* net, int32_t n_samples);

/* Byte offset of a named tensor inside the `acts` / `bwd` workspaces (for layer-by-layer parity tests and
 * debuggers); -1 if the name is unknown.  acts names: t_in c1 pool idx h0..h7 o0..o7 ds2 ds4 ds6 f8 l0 l1 q qf;
 * bwd names: dq g_l1 g_l0 g_f8 g_o0..g_o7 g_h0..g_h7 dsg2 dsg4 dsg6 g_pool g_c1, and dw:<layer> / db:<layer>
 * (f32 packed-layout weight-gradient accumulators, e.g. "dw:resnet.layer1.0.conv1"). */
int64_t vdqn_net_act_offset(const vdqn_net* net, int32_t n_samples, const char* name);
int64_t vdqn_net_bwd_offset(const vdqn_net* net, int32_t n_samples, const char* name);

/* Fold eval-mode BatchNorm into the convolutions and pack the master weights (OIHW f32) into the K-contiguous
 * forward and data-gradient operands (set_train semantics: archs/HabitatDQNMultiAction.py:37-40 — in
 * extra_capacity all 20 BatchNorm layers run on running statistics). */
int vdqn_net_pack_weights(vdqn_net* net, const float* params, const float* bnstats, void* packed, int32_t with_dgrad,
                          void* stream);
/* with_dgrad is a flag word: bit 0 = also pack the data-gradient operands, bit 1 = do NOT fold BatchNorm (the
 * convolutions then produce the raw pre-BatchNorm output; used by the train-mode BatchNorm path of 'basic'). */

/* Forward of `n_samples` samples (n_samples * F frames).  frames: see vdqn_pack_input (src_kind).
 * q_out: f32 [n_samples][num_classes*action_dim] (HabitatDQNMultiAction.forward, archs/...:44-54). */
int vdqn_net_forward(vdqn_net* net, const void* packed, const void* frames, int32_t src_kind, int32_t n_samples,
                     void* acts, float* q_out, void* stream);

/* The frozen ResNet-18 trunk only (eval-mode BatchNorm folded): frames -> 512 x 7 x 7 features, left in the `acts`
 * workspace at vdqn_net_act_offset(net, n_samples, "o7") as NHWC [n_samples*F][7][7][512].  Serves the inverse-action
 * model (archs/inverse_action2.py:50-56,72-77: `self.resnet18(k)`, `self.resnet18(k_plus_one)`), whose head is built
 * from vdqn_conv2d calls by video_dqn_amd/inverse_model.py. */
int vdqn_net_trunk_forward(vdqn_net* net, const void* packed, const void* frames, int32_t src_kind, int32_t n_samples,
                           void* acts, void* stream);

/* SyncBN for ARCHITECTURE='basic' under data parallelism (without it N ranks are not one big batch).
 * `fn` must SUM-all-reduce `count` f32 at `buf` (device memory inside the `acts` workspace passed to the forward/step
 * call) across the ranks, in place, ordered on `stream`; it is called twice per BatchNorm layer and update (forward
 * statistics, backward sums).  world_size <= 1 or fn == NULL switches it off (per-rank statistics, torch DDP's default). */
typedef void (*vdqn_allreduce_fn)(void* user, float* buf, int64_t count, void* stream);
int vdqn_net_set_bn_sync(vdqn_net* net, vdqn_allreduce_fn fn, void* user, int32_t world_size);

/* ARCHITECTURE='basic' with the module in train mode (model.train(); archs/HabitatDQNMultiAction.py:37-40 leaves the
 * ResNet's BatchNorm layers in train mode): one model call over n_samples samples with batch statistics per frame
 * slot (features is applied slot by slot, :49-51); updates `bnstats` (momentum 0.1, unbiased variance) F times per
 * BatchNorm layer.  `packed` is a workspace of vdqn_net_packed_bytes that receives the un-folded weights. */
int vdqn_net_forward_train(vdqn_net* net, const float* params, float* bnstats, void* packed, const void* frames,
                           int32_t src_kind, int32_t n_samples, void* acts, float* q_out, void* stream);

/* One TD update's device work up to the flat gradient (train_q_network.py:222-226):
 *   online forward over [before; after] (2B samples, one pass — legal because BatchNorm is in eval mode),
 *   target forward over after, fused TD loss, full backward of the `before` half.
 * Split in calls so the host can overlap the gradient all-reduce of stage s with backward stage s+1:
 *   vdqn_net_td_forward -> vdqn_net_backward_stage(0..2)  (-> all-reduce) -> vdqn_adam on the flat range. */
typedef struct vdqn_step_args {
  const float* params;        /* online master parameters (flat) */
  float* bnstats;             /* running statistics; updated in place by every update when ARCHITECTURE='basic' */
  void* packed_online;        /* workspace: vdqn_net_packed_bytes */
  const void* packed_target;  /* packed target-network weights (refreshed by the caller on target sync) */
  const void* before;         /* frames of s  : [B][F] frames, src_kind layout */
  const void* after;          /* frames of s' */
  int32_t src_kind;
  int32_t batch;              /* B (per rank) */
  const int64_t* act;
  const float* rew;
  const float* term;
  const float* valid;
  const float* gt;            /* ground-truth targets [B][5] when train_on_ground_truth, else NULL */
  float gamma;
  float inv_count;            /* 1 / (num_classes * global_batch) */
  int32_t clip_rect, linear, use_valid, train_on_ground_truth, value_learning;
  void* acts_online;          /* vdqn_net_acts_bytes(2B) (B on the ground-truth branch) */
  void* acts_target;          /* vdqn_net_acts_bytes(B); NULL on the ground-truth branch */
  void* bwd;                  /* vdqn_net_bwd_bytes(B)   */
  float* grads;               /* flat f32 [trainable_numel] */
  float* loss;                /* f32 scalar (device) */
  float* q_before;            /* optional f32 [B][15] copy of Q(s) */
  int32_t loss_kind;          /* vdqn_td_args.loss_kind (TD branch only) */
  const void* packed_frames;  /* optional: the update's frames ALREADY packed by the caller — vdqn_pack_input's output for `before`
                                 ([B*F][115][115][16]) followed by the one for `after` (TD branch), in the network's dtype, complete
                                 on `stream` when vdqn_net_td_forward is called and untouched until the update's last
                                 vdqn_net_backward_stage has run.  `before` / `after` are then not read.  For loops whose frames
                                 arrive packed (or that pack the NEXT minibatch during this update: the loader of
                                 train_q_network.py:213 has it a step ahead — measured slower on one GPU).
                                 NULL: the update packs them itself. */
  int32_t acts_samples;       /* 0: `acts_online` is laid out as vdqn_net_td_forward leaves it (2B samples, B on the
                                 ground-truth branch).  > 0: `acts_online` is the workspace of ONE vdqn_net_forward call over that
                                 many samples (== batch) — the backward of a single model call, vdqn_net_backward_begin below. */
} vdqn_step_args;
int vdqn_net_td_forward(vdqn_net* net, const vdqn_step_args* a, void* stream);
/* Stage s of the backward pass (0: head + layer4, 1: layer3, 2: layer2, layer1, stem).  With the overlap on, the stage's weight
 * gradients and the kernel that writes its range of `grads` run on the engine's side stream: after the call returns that range is
 * complete on vdqn_net_grad_stream(net), NOT on `stream`, which goes straight on to the next stage's data gradients.  The call
 * for stage 2 makes `stream` wait for the side stream, so vdqn_adam on `stream` sees every gradient.  A consumer of one stage's
 * gradients (the data-parallel all-reduce) orders itself behind vdqn_net_grad_stream. */
int vdqn_net_backward_stage(vdqn_net* net, const vdqn_step_args* a, int32_t stage, void* stream);
/* Backward of ONE earlier vdqn_net_forward(net, packed, frames, .., B, acts, ..) call from a caller-supplied dL/dQ: what
 * torch.autograd runs for `before_values = model(before)` when the reference's own loop calls loss.backward()
 * (train_q_network.py:131,226) on the HIP-backed module (video_dqn_amd/model.py).  `a` carries params, bnstats, packed_online
 * (the SAME packed weights the forward used, packed with the data-gradient operands: vdqn_net_pack_weights bit 0),
 * acts_online = that call's `acts`, batch = acts_samples = B, bwd, grads; the loss fields are not read.  dq_f32 is
 * f32 [B][num_classes*action_dim].  This call clears the weight-gradient accumulators and converts dq; the caller then runs
 * vdqn_net_backward_stage(net, a, 0..2) as after vdqn_net_td_forward.  extra_capacity only (eval-mode BatchNorm: the three
 * model calls of process_batch are independent, so each has its own backward). */
int vdqn_net_backward_begin(vdqn_net* net, const vdqn_step_args* a, const float* dq_f32, void* stream);
/* The HIP stream (hipStream_t) on which a stage's gradients become complete; NULL when the overlap is off (then it is the
 * stream passed to vdqn_net_backward_stage). */
void* vdqn_net_grad_stream(vdqn_net* net);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel exchange: one process per GPU, the flat f32 gradient SUM-all-reduced over RCCL (xGMI)
 * in the three stage buckets of vdqn_net_stage_range, each issued on vdqn_net_grad_stream(net) when its stage's
 * vdqn_net_backward_stage call has returned; the caller's stream waits for the three collectives (hipStreamWaitEvent) before
 * vdqn_adam.  The fused TD kernel already divides by the GLOBAL batch (vdqn_step_args.inv_count), so the sum is the gradient of
 * the global-batch mean loss: N ranks == the reference's one big batch (the reference itself is single-GPU,
 * train_q_network.py:255-259,275; no reference call site is replaced).
 * RCCL is looked up at run time (an already-loaded librccl.so, e.g. PyTorch's, is reused; VDQN_RCCL_LIB names another):
 * single-GPU users of this library do not need it.  The Python host uses torch.distributed's "nccl" backend — the same
 * library — instead (video_dqn_amd/dist.py); these entries serve callers that bind this header directly.
 * ------------------------------------------------------------------------------------------------ */
#define VDQN_COMM_UID_BYTES 128
typedef struct vdqn_comm vdqn_comm;
/* Rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId) and hands it to every rank by whatever channel the job has
 * (file, environment, socket); uid_host is HOST memory. */
int vdqn_comm_unique_id(void* uid_host);
/* Collective over all ranks: joins the communicator on the calling thread's current HIP device (ncclCommInitRank). */
int vdqn_comm_init(int32_t rank, int32_t nranks, const void* uid_host, vdqn_comm** out);
/* In-place SUM all-reduce of `count` elements (dtype VDQN_F32 | VDQN_BF16) at device pointer `ptr`, asynchronous on `stream`. */
int vdqn_allreduce_bucket(vdqn_comm* comm, void* ptr, int64_t count, int32_t dtype, void* stream);
int vdqn_comm_rank(const vdqn_comm* comm);
int vdqn_comm_size(const vdqn_comm* comm);
int vdqn_comm_destroy(vdqn_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* VDQN_H_ */
