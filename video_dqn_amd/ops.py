"""Operator-level Python wrappers over the C ABI (used by the per-kernel parity tests and available to
callers that want single ops).  Tensors are torch device tensors; layouts are the ABI's (NHWC activations,
[co_pad][r][s][ci] weights)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .engine import _ptr, _stream, require_gpu

TORCH_DTYPE = {_lib.VDQN_F32: torch.float32, _lib.VDQN_BF16: torch.bfloat16}


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.VDQN_F32
    if t.dtype == torch.bfloat16:
        return _lib.VDQN_BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def conv2d(x: torch.Tensor, wt: torch.Tensor, *, ho: int, wo: int, co: int, r: int, s: int, stride: int, pad: int,
           bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
           relu: bool = False, mode: int = 0, pix_stride: Optional[int] = None, ci: Optional[int] = None,
           want_f32: bool = False, ldo: Optional[int] = None, want_colsum: bool = False,
           wt2: Optional[torch.Tensor] = None, bias2: Optional[torch.Tensor] = None, co2: int = 0, relu2: bool = False,
           in2: Optional[torch.Tensor] = None):
    """x: [n, hi, wi, c] NHWC; wt: [co_pad, r, s, ci].  Returns out [n, ho, wo, ldo] (and the f32 copy).
    Fused sibling 1x1 / stride 2 (include/vdqn.h): forward — wt2 [co2, 1, 1, ci] (+ bias2, relu2) gives a second output
    (returned after `out`); data gradient — in2 [n, hi, wi, ci2] and wt2 [co, 1, 1, ci2] add the sibling's gradient."""
    lib = _lib.load()
    require_gpu()
    n, hi, wi, cx = x.shape
    ci = cx if ci is None else ci
    pix_stride = cx if pix_stride is None else pix_stride
    ldo = co if ldo is None else ldo
    out = torch.empty((n, ho, wo, ldo), dtype=x.dtype, device=x.device)
    out_f32 = torch.empty((n, ho, wo, ldo), dtype=torch.float32, device=x.device) if want_f32 else None
    a = _lib.ConvArgs()
    a.in_, a.wt, a.bias, a.resid, a.mask = _ptr(x), _ptr(wt), _ptr(bias), _ptr(resid), _ptr(mask)
    a.out, a.out_f32 = _ptr(out), _ptr(out_f32)
    a.n_img, a.hi, a.wi, a.ci, a.pix_stride = n, hi, wi, ci, pix_stride
    a.ho, a.wo, a.co, a.ldo = ho, wo, co, ldo
    a.r, a.s, a.stride, a.pad = r, s, stride, pad
    a.mode, a.relu, a.dtype = mode, int(relu), dtype_code(x)
    # every field that decides which kernel takes the call is set BEFORE the colsum-row query (the skinny kernels refuse a call with
    # column sums or a sibling: a query on half-filled args would answer for another kernel than the one that runs)
    out2 = None
    if wt2 is not None and mode == 0:
        out2 = torch.empty((n, ho, wo, co2), dtype=x.dtype, device=x.device)
        a.wt2, a.bias2, a.out2, a.co2, a.ldo2, a.relu2 = _ptr(wt2), _ptr(bias2), _ptr(out2), co2, co2, int(relu2)
    elif wt2 is not None:
        a.wt2, a.in2, a.ci2 = _ptr(wt2), _ptr(in2), in2.shape[-1]
    part = None
    if want_colsum:
        a.colsum_part = 16  # non-null placeholder for the query; the real buffer is set below
        rows = lib.vdqn_conv2d_colsum_rows(C.byref(a))  # 128, or the row tile of the kernel that takes this call
        if mode == 1 and stride == 2:  # one run of tiles per output-parity class
            n_tiles = sum((n * hc * wc + rows - 1) // rows for hc in ((ho + 1) // 2, ho // 2) for wc in ((wo + 1) // 2, wo // 2))
        else:
            n_tiles = (n * ho * wo + rows - 1) // rows
        part = torch.empty((n_tiles, ldo), dtype=torch.float32, device=x.device)
    a.colsum_part = _ptr(part)
    _lib.check(lib.vdqn_conv2d(C.byref(a), _stream()), "vdqn_conv2d")
    if out2 is not None:
        return out, out2
    if want_colsum:
        return out, part
    return (out, out_f32) if want_f32 else out


def conv2d_wgrad(gy: torch.Tensor, x: torch.Tensor, *, co: int, r: int, s: int, stride: int, pad: int,
                 ci: Optional[int] = None, pix_stride: Optional[int] = None, splitk: int = 0, want_dbias: bool = True,
                 deterministic: bool = False, poison_workspace: bool = False):
    """gy: [n, ho, wo, ldg]; x: [n, hi, wi, c].  Returns dw f32 [co_pad, r, s, ci] (and dbias [co_pad]).
    deterministic: the two-stage ordered reduction (a workspace of vdqn_conv2d_wgrad_workspace_bytes) instead of atomics;
    poison_workspace fills that workspace with NaN first (tests: nothing the call does not write may reach dw)."""
    lib = _lib.load()
    n, ho, wo, ldg = gy.shape
    _, hi, wi, cx = x.shape
    ci = cx if ci is None else ci
    pix_stride = cx if pix_stride is None else pix_stride
    co_pad = (co + 63) // 64 * 64
    dw = torch.zeros((co_pad, r, s, ci), dtype=torch.float32, device=x.device)
    db = torch.zeros((co_pad,), dtype=torch.float32, device=x.device) if want_dbias else None
    a = _lib.WgradArgs()
    a.gy, a.x, a.dw, a.dbias = _ptr(gy), _ptr(x), _ptr(dw), _ptr(db)
    a.n_img, a.hi, a.wi, a.ci, a.pix_stride = n, hi, wi, ci, pix_stride
    a.ho, a.wo, a.co, a.ldg = ho, wo, co, ldg
    a.r, a.s, a.stride, a.pad = r, s, stride, pad
    a.splitk, a.dtype = splitk, dtype_code(x)
    ws = None
    if deterministic:
        nbytes = lib.vdqn_conv2d_wgrad_workspace_bytes(C.byref(a))
        if nbytes < 0:
            _lib.check(-1, "vdqn_conv2d_wgrad_workspace_bytes")
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
        if poison_workspace:
            ws.fill_(0xFF)  # every f32 word a NaN
        a.workspace, a.workspace_bytes = _ptr(ws), nbytes
    _lib.check(lib.vdqn_conv2d_wgrad(C.byref(a), _stream()), "vdqn_conv2d_wgrad")
    return (dw, db) if want_dbias else dw


def pack_input(src: torch.Tensor, src_kind: int, n_img: int, dtype: torch.dtype) -> torch.Tensor:
    lib = _lib.load()
    dst = torch.empty((n_img, 115, 115, 16), dtype=dtype, device=src.device)
    _lib.check(lib.vdqn_pack_input(_ptr(src), src_kind, _ptr(dst), n_img, dtype_code(dst), _stream()), "vdqn_pack_input")
    return dst


def maxpool_fwd(x: torch.Tensor):
    lib = _lib.load()
    n, hi, wi, c = x.shape
    ho, wo = (hi - 1) // 2 + 1, (wi - 1) // 2 + 1
    out = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    _lib.check(lib.vdqn_maxpool_fwd(_ptr(x), _ptr(out), _ptr(idx), n, hi, wi, c, dtype_code(x), _stream()), "vdqn_maxpool_fwd")
    return out, idx


def maxpool_bwd(gy: torch.Tensor, idx: torch.Tensor, x: Optional[torch.Tensor] = None, out_hw=None) -> torch.Tensor:
    """x given: the gradient is also masked by x > 0 (the ReLU in front of the pool); x None: plain unpooling into out_hw."""
    lib = _lib.load()
    if x is not None:
        n, hi, wi, c = x.shape
        gx = torch.empty_like(x)
    else:
        n, _, _, c = gy.shape
        hi, wi = out_hw
        gx = torch.empty((n, hi, wi, c), dtype=gy.dtype, device=gy.device)
    _lib.check(lib.vdqn_maxpool_bwd(_ptr(gy), _ptr(idx), _ptr(x), _ptr(gx), n, hi, wi, c, dtype_code(gy), _stream()), "vdqn_maxpool_bwd")
    return gx


def stem_wgrad_pool(g_pool: torch.Tensor, idx: torch.Tensor, t_in: torch.Tensor, deterministic: bool = False) -> torch.Tensor:
    """conv1's weight gradient from the pooled gradient: g_pool [n,56,56,64] bf16, idx uint8 (stem_conv_pool / maxpool_fwd),
    t_in [n,115,115,16] (pack_input) -> dw f32 [64,4,1,64] == conv2d_wgrad(maxpool_bwd(g_pool, idx), t_in) on the stem geometry."""
    lib = _lib.load()
    n = g_pool.shape[0]
    dw = torch.zeros((64, 4, 1, 64), dtype=torch.float32, device=g_pool.device)
    ws, nbytes = None, 0
    if deterministic:
        nbytes = lib.vdqn_stem_wgrad_pool_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g_pool.device)
    _lib.check(lib.vdqn_stem_wgrad_pool(_ptr(g_pool), _ptr(idx), _ptr(t_in), _ptr(dw), n, _ptr(ws), nbytes, _stream()), "vdqn_stem_wgrad_pool")
    return dw


LOSS_KINDS = {"l2": 0, "huber": 1}  # vdqn_td_args.loss_kind


def td_loss(q_before, q_after_online, q_after_target, act, rew, term, valid=None, *, n_cat=5, n_act=3, gamma=0.99,
            inv_count=None, clip_rect=True, linear=False, out_dtype=torch.float32, loss_kind="l2"):
    """q_*: f32 [B, ldq] (ldq >= n_cat*n_act).  Returns (loss[1], dq[B, ldq] out_dtype, dq_f32)."""
    lib = _lib.load()
    B, ldq = q_before.shape
    dev = q_before.device
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    dq = torch.empty((B, ldq), dtype=out_dtype, device=dev)
    dq32 = torch.empty((B, ldq), dtype=torch.float32, device=dev)
    a = _lib.TdArgs()
    a.q_before, a.q_after_online, a.q_after_target = _ptr(q_before), _ptr(q_after_online), _ptr(q_after_target)
    a.act, a.rew, a.term, a.valid = _ptr(act), _ptr(rew), _ptr(term), _ptr(valid)
    a.loss, a.dq, a.dq_f32 = _ptr(loss), _ptr(dq), _ptr(dq32)
    a.batch, a.n_cat, a.n_act, a.ldq = B, n_cat, n_act, ldq
    a.gamma = gamma
    a.inv_count = (1.0 / (B * n_cat)) if inv_count is None else inv_count
    a.clip_rect, a.linear, a.use_valid, a.dtype = int(clip_rect), int(linear), int(valid is not None), dtype_code(dq)
    a.loss_kind = LOSS_KINDS[loss_kind]
    _lib.check(lib.vdqn_td_loss(C.byref(a), _stream()), "vdqn_td_loss")
    return loss, dq, dq32


def gt_loss(q_before, act, gt, *, n_cat=5, n_act=3, inv_count=None, value_learning=False):
    lib = _lib.load()
    B, ldq = q_before.shape
    dev = q_before.device
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    dq32 = torch.empty((B, ldq), dtype=torch.float32, device=dev)
    inv = (1.0 / (B * n_cat)) if inv_count is None else inv_count
    _lib.check(lib.vdqn_gt_loss(_ptr(q_before), _ptr(act), _ptr(gt), _ptr(loss), None, _ptr(dq32), B, n_cat, n_act, ldq,
                                inv, int(value_learning), _lib.VDQN_F32, _stream()), "vdqn_gt_loss")
    return loss, dq32


def adam(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    lib = _lib.load()
    _lib.check(lib.vdqn_adam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), step, lr, beta1, beta2, eps, _stream()), "vdqn_adam")


def bn_train_fwd(y: torch.Tensor, gamma, beta, running_mean=None, running_var=None, *, resid=None, relu=False,
                 num_frames=1, imgs_per_half=None, momentum=0.1, eps=1e-5, deterministic=False):
    """Train-mode BatchNorm2d over NHWC y [n, h, w, c] (statistic groups: see include/vdqn.h).
    Returns (z, work) — `work` f32 [groups, 6, c] is what bn_train_bwd needs.
    deterministic: ordered two-stage statistic sums (a NaN-filled workspace of vdqn_bn_train_workspace_bytes) instead of atomics."""
    lib = _lib.load()
    n, h, w, c = y.shape
    iph = n if imgs_per_half is None else imgs_per_half
    groups = n // iph * num_frames
    z = torch.empty_like(y)
    work = torch.zeros((groups, 6, c), dtype=torch.float32, device=y.device)
    ws, nbytes = _bn_workspace(lib, y, n, h * w, c, num_frames, iph, deterministic)
    _lib.check(lib.vdqn_bn_train_fwd(_ptr(y), _ptr(resid), _ptr(z), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                     _ptr(work), n, h * w, c, num_frames, iph, int(relu), momentum, eps, dtype_code(y), _ptr(ws), nbytes,
                                     _stream()), "vdqn_bn_train_fwd")
    return z, work


def _bn_workspace(lib, y, n, hw, c, num_frames, iph, deterministic):
    if not deterministic:
        return None, 0
    nbytes = lib.vdqn_bn_train_workspace_bytes(n, hw, c, num_frames, iph)
    if nbytes < 0:
        _lib.check(-1, "vdqn_bn_train_workspace_bytes")
    return torch.full((max(nbytes, 16),), 0xFF, dtype=torch.uint8, device=y.device), nbytes


def bn_train_bwd(g: torch.Tensor, y: torch.Tensor, work: torch.Tensor, *, num_frames=1, imgs_per_half=None, deterministic=False):
    """Returns (dy, dgamma, dbeta) for the BatchNorm whose forward filled `work`."""
    lib = _lib.load()
    n, h, w, c = y.shape
    iph = n if imgs_per_half is None else imgs_per_half
    dy = torch.empty_like(y)
    dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
    ws, nbytes = _bn_workspace(lib, y, n, h * w, c, num_frames, iph, deterministic)
    _lib.check(lib.vdqn_bn_train_bwd(_ptr(g), _ptr(y), _ptr(dy), _ptr(work), _ptr(dgamma), _ptr(dbeta), n, h * w, c, num_frames, iph,
                                     dtype_code(y), _ptr(ws), nbytes, _stream()), "vdqn_bn_train_bwd")
    return dy, dgamma, dbeta


def avgpool_fwd(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    n, h, w, c = x.shape
    out = torch.empty((n, c), dtype=x.dtype, device=x.device)
    _lib.check(lib.vdqn_avgpool_fwd(_ptr(x), _ptr(out), n, h * w, c, dtype_code(x), _stream()), "vdqn_avgpool_fwd")
    return out


def avgpool_bwd(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g [n, c]; x [n, h, w, c] (post-ReLU input of the pool): gx = (x > 0) * g / (h*w)."""
    lib = _lib.load()
    n, h, w, c = x.shape
    gx = torch.empty_like(x)
    _lib.check(lib.vdqn_avgpool_bwd(_ptr(g), _ptr(x), _ptr(gx), n, h * w, c, dtype_code(x), _stream()), "vdqn_avgpool_bwd")
    return gx


def stem_conv_pool(t_in: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor, want_idx: bool = True, n_idx: int = None):
    """t_in [n,115,115,16] (pack_input), wt [64,4,1,64], bias f32 [64] -> (pool [n,56,56,64], idx uint8).

    want_idx=False: frames that never see a backward pass — idx is None and the arg-max bytes are not computed.
    n_idx: arg-max bytes for the first n_idx images only (vdqn_stem_conv_pool_n); the other rows of idx stay as allocated."""
    lib = _lib.load()
    n = t_in.shape[0]
    pool = torch.empty((n, 56, 56, 64), dtype=t_in.dtype, device=t_in.device)
    if n_idx is not None:
        idx = torch.zeros((n, 56, 56, 64), dtype=torch.uint8, device=t_in.device)
        _lib.check(lib.vdqn_stem_conv_pool_n(_ptr(t_in), _ptr(wt), _ptr(bias), _ptr(pool), _ptr(idx), n, int(n_idx), dtype_code(t_in), _stream()),
                   "vdqn_stem_conv_pool_n")
        return pool, idx
    idx = torch.empty((n, 56, 56, 64), dtype=torch.uint8, device=t_in.device) if want_idx else None
    _lib.check(lib.vdqn_stem_conv_pool(_ptr(t_in), _ptr(wt), _ptr(bias), _ptr(pool), _ptr(idx) if want_idx else None, n, dtype_code(t_in), _stream()),
               "vdqn_stem_conv_pool")
    return pool, idx
