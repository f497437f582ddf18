"""Training of the inverse-action model on the HIP kernels (``train_inverse_model.py:30-200``).

The frozen ResNet-18 trunk runs through ``vdqn_net_trunk_forward`` (no gradient reaches it, :41-44); the head
(conv 1x1 -> 3x3 -> 3x3 -> fc1 -> dropout -> fc2 -> ReLU -> fc_accuracy, the TRAINING script's variant with the ReLU after
fc2, :66-82) is a chain of ``vdqn_conv2d`` / ``vdqn_conv2d_wgrad`` calls; loss = ``vdqn_softmax_ce`` (nn.CrossEntropyLoss,
:103-104); optimiser = ``vdqn_adam`` (+ ``vdqn_axpy`` for weight_decay, :190); ``StepLR`` (:193,199) is a host scalar.
torch is used for device memory, for re-laying-out the small head weights/gradients between OIHW and the kernels'
K-contiguous layout, and for drawing the dropout mask when the caller does not supply one."""
from __future__ import annotations

import torch

from . import _lib, ops
from .engine import _ptr, _stream
from .inverse_model import InverseActionModel

HEAD = ("conv1", "conv2", "conv3", "fc1", "fc2", "fc_accuracy")


def _conv_w(w, dt):  # OIHW f32 -> [co_pad][r][s][ci]
    co_pad = (w.shape[0] + 63) // 64 * 64
    out = torch.zeros((co_pad, w.shape[2], w.shape[3], w.shape[1]), dtype=dt, device=w.device)
    out[:w.shape[0]] = w.permute(0, 2, 3, 1).to(dt)
    return out.contiguous()


def _conv_wd(w, dt):  # OIHW f32 -> data-gradient operand [ci][r][s][co_pad]
    co_pad = (w.shape[0] + 63) // 64 * 64
    out = torch.zeros((w.shape[1], w.shape[2], w.shape[3], co_pad), dtype=dt, device=w.device)
    out[..., :w.shape[0]] = w.permute(1, 2, 3, 0).to(dt)
    return out.contiguous()


def _pad_bias(b):
    out = torch.zeros(((b.numel() + 63) // 64 * 64,), dtype=torch.float32, device=b.device)
    out[:b.numel()] = b
    return out


class InverseTrainer:
    def __init__(self, model: InverseActionModel, lr=1e-4, weight_decay=0.0, lr_decay=0.1, lr_decay_every=200, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr0, self.wd, self.betas, self.eps = model, lr, weight_decay, betas, eps
        self.lr_decay, self.lr_decay_every = lr_decay, lr_decay_every
        self.epoch = 0
        dev = model.engine.device
        self.names = [n + sfx for n in HEAD for sfx in (".weight", ".bias")]
        self.numel = [model.head[n].numel() for n in self.names]
        total = sum(self.numel)
        # flat master copy of the head (Adam runs on one range); model.head tensors become views into it
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, self.gviews = {}, {}
        off = 0
        for n, k in zip(self.names, self.numel):
            self.flat[off:off + k].copy_(model.head[n].reshape(-1))
            self.views[n] = self.flat[off:off + k].view(model.head[n].shape)
            self.gviews[n] = self.grad[off:off + k].view(model.head[n].shape)
            model.head[n] = self.views[n]
            off += k
        self.step_count = 0
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)

    @property
    def lr(self):  # StepLR(step_size=lr_decay_every, gamma=lr_decay), stepped once per epoch (train_inverse_model.py:193,199)
        return self.lr0 * self.lr_decay ** (self.epoch // int(self.lr_decay_every))

    def end_epoch(self):
        self.epoch += 1

    def step(self, k, k_plus_one, act, dropout_mask=None):
        """One minibatch of train_inverse_model.py:93-112.  Returns (device loss scalar, y [B,3])."""
        mdl, eng = self.model, self.model.engine
        dt = mdl.tdtype
        code = ops.dtype_code(torch.empty(0, dtype=dt))
        B = k.shape[0]
        src_kind = 0 if k.dtype == torch.uint8 else 1
        frames = (torch.cat([k, k_plus_one], 0) if src_kind == 0 else torch.cat([k.float(), k_plus_one.float()], 0)).to(eng.device).contiguous()
        h = self.views
        with torch.cuda.device(eng.device):
            if eng._packed_version != eng.version_key():
                eng.pack_weights()
                eng._packed_version = eng.version_key()
            acts = eng._acts_for(2 * B)
            _lib.check(eng.lib.vdqn_net_trunk_forward(eng.handle, _ptr(eng.packed), _ptr(frames), src_kind, 2 * B, _ptr(acts), _stream()),
                       "vdqn_net_trunk_forward")
            off = eng.lib.vdqn_net_act_offset(eng.handle, 2 * B, b"o7")
            esz = 2 if dt == torch.bfloat16 else 4
            feat = acts[off:off + 2 * B * 49 * 512 * esz].view(dt).view(2 * B, 7, 7, 512)
            x0 = torch.cat([feat[:B], feat[B:]], dim=3).contiguous()
            # ---- forward (train mode) ----
            w1, w2, w3 = _conv_w(h["conv1.weight"], dt), _conv_w(h["conv2.weight"], dt), _conv_w(h["conv3.weight"], dt)
            a1 = ops.conv2d(x0, w1, ho=7, wo=7, co=256, r=1, s=1, stride=1, pad=0, bias=_pad_bias(h["conv1.bias"]), relu=True)
            a2 = ops.conv2d(a1, w2, ho=5, wo=5, co=256, r=3, s=3, stride=1, pad=0, bias=_pad_bias(h["conv2.bias"]), relu=True)
            a3 = ops.conv2d(a2, w3, ho=3, wo=3, co=64, r=3, s=3, stride=1, pad=0, bias=_pad_bias(h["conv3.bias"]), relu=True)
            wf1 = h["fc1.weight"].view(128, 64, 9).permute(0, 2, 1).reshape(128, 576, 1, 1)  # NCHW flatten -> NHWC flatten
            a3f = a3.view(B, 1, 1, 576)
            h1 = ops.conv2d(a3f, _conv_w(wf1, dt), ho=1, wo=1, co=128, r=1, s=1, stride=1, pad=0, bias=_pad_bias(h["fc1.bias"]), relu=True)
            if dropout_mask is None:
                dropout_mask = (torch.rand((B, 128), device=eng.device) >= 0.5)
            mask = dropout_mask.to(eng.device).to(dt).view(B, 1, 1, 128).contiguous()
            d1 = torch.empty_like(h1)
            _lib.check(eng.lib.vdqn_mask_scale(_ptr(h1), _ptr(mask), _ptr(d1), h1.numel(), 2.0, code, _stream()), "vdqn_mask_scale")
            w_fc2 = torch.zeros((64, 128, 1, 1), device=eng.device)
            w_fc2[:3] = h["fc2.weight"].view(3, 128, 1, 1)
            h2 = ops.conv2d(d1, _conv_w(w_fc2, dt), ho=1, wo=1, co=64, r=1, s=1, stride=1, pad=0, bias=_pad_bias(h["fc2.bias"]), relu=True)
            w_acc = torch.zeros((64, 64, 1, 1), device=eng.device)
            w_acc[:3, :3] = h["fc_accuracy.weight"].view(3, 3, 1, 1)
            _, y32 = ops.conv2d(h2, _conv_w(w_acc, dt), ho=1, wo=1, co=64, r=1, s=1, stride=1, pad=0, bias=_pad_bias(h["fc_accuracy.bias"]), want_f32=True)
            y32 = y32.view(B, 64)
            # ---- loss + backward of the head ----
            self.loss.zero_()
            dy = torch.empty((B, 1, 1, 64), dtype=dt, device=eng.device)
            labels = act.to(eng.device).to(torch.int64).contiguous()
            _lib.check(eng.lib.vdqn_softmax_ce(_ptr(y32), _ptr(labels), _ptr(self.loss), _ptr(dy), B, 64, 3, 1.0 / B, code, _stream()), "vdqn_softmax_ce")

            def wgrad(gy, x, co, r):
                dw, db = ops.conv2d_wgrad(gy, x, co=co, r=r, s=r, stride=1, pad=0)
                return dw, db

            def dgrad(gy, w_oihw, hi, ci, r, mask_t=None):
                return ops.conv2d(gy, _conv_wd(w_oihw, dt), ho=hi, wo=hi, co=ci, r=r, s=r, stride=1, pad=0, mode=1, mask=mask_t)
            g = self.gviews
            dw, db = wgrad(dy, h2, 64, 1)
            g["fc_accuracy.weight"].copy_(dw[:3, 0, 0, :3]); g["fc_accuracy.bias"].copy_(db[:3])
            g_h2 = dgrad(dy, w_acc, 1, 64, 1, h2)                     # through fc_accuracy and the ReLU after fc2
            dw, db = wgrad(g_h2, d1, 64, 1)
            g["fc2.weight"].copy_(dw[:3, 0, 0, :]); g["fc2.bias"].copy_(db[:3])
            g_d1 = dgrad(g_h2, w_fc2, 1, 128, 1, h1)                  # ReLU of fc1 (commutes with the dropout mask)
            g_h1 = torch.empty_like(g_d1)
            _lib.check(eng.lib.vdqn_mask_scale(_ptr(g_d1), _ptr(mask), _ptr(g_h1), g_d1.numel(), 2.0, code, _stream()), "vdqn_mask_scale")
            dw, db = wgrad(g_h1, a3f, 128, 1)
            g["fc1.weight"].copy_(dw[:128, 0, 0, :].view(128, 9, 64).permute(0, 2, 1).reshape(128, 576)); g["fc1.bias"].copy_(db[:128])
            g_a3 = dgrad(g_h1, wf1, 1, 576, 1, a3f).view(B, 3, 3, 64)
            dw, db = wgrad(g_a3, a2, 64, 3)
            g["conv3.weight"].copy_(dw[:64].permute(0, 3, 1, 2)); g["conv3.bias"].copy_(db[:64])
            g_a2 = dgrad(g_a3, h["conv3.weight"], 5, 256, 3, a2)
            dw, db = wgrad(g_a2, a1, 256, 3)
            g["conv2.weight"].copy_(dw[:256].permute(0, 3, 1, 2)); g["conv2.bias"].copy_(db[:256])
            g_a1 = dgrad(g_a2, h["conv2.weight"], 7, 256, 3, a1)
            dw, db = wgrad(g_a1, x0, 256, 1)
            g["conv1.weight"].copy_(dw[:256].permute(0, 3, 1, 2)); g["conv1.bias"].copy_(db[:256])
            # ---- Adam (weight_decay folded into the gradient as torch does) ----
            if self.wd:
                _lib.check(eng.lib.vdqn_axpy(_ptr(self.grad), _ptr(self.flat), float(self.wd), self.flat.numel(), _stream()), "vdqn_axpy")
            self.step_count += 1
            _lib.check(eng.lib.vdqn_adam(_ptr(self.flat), _ptr(self.grad), _ptr(self.m), _ptr(self.v), self.flat.numel(), self.step_count,
                                         self.lr, self.betas[0], self.betas[1], self.eps, _stream()), "vdqn_adam")
            mdl._packed_head = None
        return self.loss, y32[:, :3].clone()
