"""A bf16-EMULATING form of the CPU oracle.  TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

The throughput mode of the engine stores activations, gradients and packed weights in bf16 and accumulates in f32.  Against the
plain fp32 oracle its gradients can only be held to cosine / norm gates (thousands of ReLU decisions differ).  This module
evaluates the SAME network as `oracle/ref_cpu.py` (archs/HabitatDQNMultiAction.py:44-54 over the torchvision ResNet-18 topology;
the loss stays `ref_cpu.process_batch`, train_q_network.py:126-181) with a rounding to bf16 at every point where the engine
rounds, and nowhere else, so that the bf16 kernels the benchmark times (`win9u`, `win9s`, `conv64`, `stem_kernel`, `wgrad_win`,
`stem_wgrad_pool`) meet an element-wise check inside a real update:

  forward   the normalised frame (vdqn_pack_input), the BatchNorm-folded weights W * gamma * rstd of every convolution and the
            weights of `features.8` / `top.*` (fold kernels of engine.hip), every stored activation (conv epilogue: f32 accumulator
            + f32 bias (+ residual) -> ReLU -> bf16), the downsample branch's output; Q itself stays f32 (`qf`);
  backward  every stored gradient: dL/dQ (td_loss writes dq in bf16), the gradient with respect to every pre-ReLU sum (what a data
            gradient kernel stores: (dgrad + residual path) * mask -> bf16), the pooled gradient and the max-pool backward's
            output; weight gradients accumulate bf16 x bf16 products in f32 and are never rounded; the BatchNorm fold's chain
            rule (dW = dW' * s, dbeta = column sums of the stored gradient, dgamma through both) is left to autograd.

Rounding is torch's f32 -> bf16 conversion (round to nearest even), the same rule as v_cvt_pk_bf16_f32.  With `relu_masks` (the
ENGINE's stored ReLU decisions of the model(before) pass, tests/test_gpu_engine.py `_engine_relu_masks`) the first call
differentiates the piecewise-linear function the engine evaluated."""
import torch
import torch.nn.functional as F


ROUNDING = True  # False: every rounding point is the identity (the structure check of tests/test_bf16_emulation_cpu.py)


def bf16r(x):
    return x.to(torch.bfloat16).to(x.dtype) if ROUNDING else x


class _Round(torch.autograd.Function):
    """y = bf16(x) if fwd else x;  dx = bf16(dy) if bwd else dy."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return bf16r(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (bf16r(g) if ctx.bwd else g), None, None


def q_act(x):      # a stored activation and the stored gradient with respect to it
    return _Round.apply(x, True, True)


def q_fwd(x):      # stored activation whose gradient is not stored separately (the downsample branch: it IS the block's gradient)
    return _Round.apply(x, True, False)


def q_bwd(x):      # f32 forward value with a bf16 gradient (Q itself; the pooled map, already bf16)
    return _Round.apply(x, False, True)


def q_weight(w):   # packed bf16 operand of an f32 master weight: straight-through (the fold's gradient is dW' * s in f32)
    return w + (bf16r(w) - w).detach()


def _conv_bn(x, conv, bn, stride, pad):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = q_weight(conv.weight * s.view(-1, 1, 1, 1))
    b = bn.bias - bn.running_mean * s
    return F.conv2d(x, w, None, stride, pad) + b.view(1, -1, 1, 1)


def _relu(z, mask, grec=None, name=None):
    """ReLU of a pre-activation that is stored in bf16: value mask * bf16(z), gradient bf16(mask * g)."""
    if grec is not None and z.requires_grad:
        z.register_hook(lambda g, n=name: grec.setdefault(n, []).append(g.detach().clone()))
    zq = q_act(z)
    return zq * mask.to(zq.dtype) if mask is not None else torch.relu(zq)


class EmulatedNet(torch.nn.Module):
    """Callable like the oracle's HabitatDQNMultiAction `m` (extra_capacity, eval-mode BatchNorm), computing with the engine's
    bf16 rounding points; gradients land in `m`'s own parameters.  `relu_masks[b]` = the engine's ReLU decisions of BasicBlock b
    in call order (h, o per frame slot) for the FIRST call only; `record` (dict) receives that call's stored activations."""

    def __init__(self, m, relu_masks=None, record=None, graph_first_call_only=False, grad_record=None, head_masks=None):
        super().__init__()
        self.m, self.relu_masks, self.record, self.calls = m, relu_masks, record, 0
        # head_masks: the engine's ReLU decisions of the head for the first call — {"f8": [per frame slot, NCHW bool], "l0": [B, 512],
        # "l1": [B, 256]}.  Needed as much as the blocks' masks: bf16 activations carry ~5e-3 of noise that no emulation can track bit
        # for bit through 17 layers (a different f32 summation order flips roundings, and the flips compound), so ~1 % of the head's
        # units sit on the other side of zero — each one a full-size error in the gradient below it.
        self.head_masks = head_masks
        # grad_record (dict): the first call's STORED gradients, as the engine keeps them (rounded, masked): "dq", "g_l1", "g_l0",
        # lists per frame slot "g_f8", "g_o<b>", "g_h<b>", "g_pool" — filled during backward
        self.grad_record = grad_record
        # process_batch differentiates model(before) only (train_q_network.py:131,226); model(after) feeds an argmax.  At the
        # benchmark's batch the graph of that second call is ~10 GB of host memory for nothing: True runs it under no_grad
        self.graph_first_call_only = graph_first_call_only

    def set_train(self):
        self.m.set_train()

    def _features(self, x, slot, masks, rec, grec=None, hm=None):
        r = self.m.resnet
        x = bf16r(x)
        a1 = torch.relu(q_act(_conv_bn(x, r.conv1, r.bn1, 2, 3)))
        pm = F.max_pool2d(a1, 3, 2, 1)
        if grec is not None and pm.requires_grad:
            pm.register_hook(lambda g: grec.setdefault("g_pool", []).append(g.detach().clone()))
        p = q_bwd(pm)
        if rec is not None:
            rec.setdefault("pool", []).append(p.detach())
        xin = p
        for b in range(8):
            blk = getattr(r, f"layer{b // 2 + 1}")[b % 2]
            stride = 2 if (b % 2 == 0 and b > 0) else 1
            mh = masks[b][2 * slot] if masks is not None else None
            mo = masks[b][2 * slot + 1] if masks is not None else None
            h = _relu(_conv_bn(xin, blk.conv1, blk.bn1, stride, 1), mh, grec, f"g_h{b}")
            identity = xin
            if blk.downsample is not None:
                identity = q_fwd(_conv_bn(xin, blk.downsample[0], blk.downsample[1], stride, 0))
                if rec is not None:
                    rec.setdefault(f"ds{b}", []).append(identity.detach())
            o = _relu(_conv_bn(h, blk.conv2, blk.bn2, 1, 1) + identity, mo, grec, f"g_o{b}")
            if rec is not None:
                rec.setdefault(f"h{b}", []).append(h.detach())
                rec.setdefault(f"o{b}", []).append(o.detach())
            xin = o
        f8 = self.m.features[8]
        z = F.conv2d(xin, q_weight(f8.weight), None) + f8.bias.view(1, -1, 1, 1)
        f = _relu(z, hm["f8"][slot] if hm is not None else None, grec, "g_f8")
        if rec is not None:
            rec.setdefault("f8", []).append(f.detach())
        return torch.flatten(f, 1)

    def forward(self, inp):
        if self.graph_first_call_only and self.calls > 0 and torch.is_grad_enabled():
            with torch.no_grad():
                return self.forward(inp)
        m = self.m
        first = self.calls == 0
        self.calls += 1
        masks = self.relu_masks if first else None
        rec = self.record if first else None
        grec = self.grad_record if first else None
        hm = self.head_masks if first else None
        if m.num_frames == 1 and inp.dim() == 4:
            inp = inp.unsqueeze(1)
        if inp.shape[1] != m.num_frames:
            raise Exception("bad shape")
        feats = [self._features(inp[:, i], i, masks, rec, grec, hm) for i in range(m.num_frames)]
        x = torch.cat(feats, 1)
        l0 = _relu(F.linear(x, q_weight(m.top[0].weight)) + m.top[0].bias, hm["l0"] if hm is not None else None, grec, "g_l0")
        l1 = _relu(F.linear(l0, q_weight(m.top[2].weight)) + m.top[2].bias, hm["l1"] if hm is not None else None, grec, "g_l1")
        q = q_bwd(F.linear(l1, q_weight(m.top[4].weight)) + m.top[4].bias)
        if grec is not None and q.requires_grad:
            q.register_hook(lambda g: grec.setdefault("dq_f32", []).append(g.detach().clone()))
        if rec is not None:
            rec["l0"], rec["l1"], rec["q"] = l0.detach(), l1.detach(), q.detach()
        return q.view((-1, m.num_classes, m.action_dim))
