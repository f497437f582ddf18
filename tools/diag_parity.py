#!/usr/bin/env python
"""Diagnostic (GPU box): actual error numbers of the HIP path vs the CPU oracle, per op and per parameter."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_cpu  # noqa: E402
from video_dqn_amd import ops, synth  # noqa: E402
from video_dqn_amd.engine import NetEngine, TDStepper  # noqa: E402

DEV = "cuda"


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def cos(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def rnd(seed, name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(synth.uniform(seed, name, shape, lo, hi))


def op_errors():
    for dtype in (torch.float32, torch.bfloat16):
        n, ci, co, h, k, stride, pad = 2, 128, 128, 14, 3, 1, 1
        x = rnd(1, "x", (n, ci, h, h)).to(dtype).float()
        w = rnd(2, "w", (co, ci, k, k), -0.1, 0.1).to(dtype).float()
        gy = rnd(5, "gy", (n, co, h, h)).to(dtype).float()
        xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
        gyd = gy.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
        wf = w.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
        wd = w.permute(1, 2, 3, 0).contiguous().to(dtype).to(DEV)
        ref64 = F.conv2d(x.double(), w.double(), None, stride, pad)
        _, o32 = ops.conv2d(xd, wf, ho=h, wo=h, co=co, r=k, s=k, stride=stride, pad=pad, want_f32=True)
        print(dtype, "conv fwd  gpu-vs-f64 %.2e   cpu32-vs-f64 %.2e" % (
            relerr(o32.cpu().permute(0, 3, 1, 2), ref64), relerr(F.conv2d(x, w, None, stride, pad), ref64)))
        dref = F.grad.conv2d_input((n, ci, h, h), w.double(), gy.double(), stride, pad)
        g = ops.conv2d(gyd, wd, ho=h, wo=h, co=ci, r=k, s=k, stride=stride, pad=pad, mode=1)
        print(dtype, "conv dgrad gpu-vs-f64 %.2e" % relerr(g.float().cpu().permute(0, 3, 1, 2), dref))
        wref = F.grad.conv2d_weight(x.double(), (co, ci, k, k), gy.double(), stride, pad)
        dw = ops.conv2d_wgrad(gyd, xd, co=co, r=k, s=k, stride=stride, pad=pad, want_dbias=False)
        print(dtype, "conv wgrad gpu-vs-f64 %.2e" % relerr(dw.cpu().permute(0, 3, 1, 2), wref))


def step_errors(dtype, B=8):
    net = NetEngine(3, 5, 1, True, dtype, 2 * B)
    net.load_tensors(synth.make_state_dict(7))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    tnet = NetEngine(3, 5, 1, True, dtype, 2 * B)
    tnet.load_tensors(synth.make_state_dict(8))
    tnet.pack_weights(stp.packed_target)
    (tup, raw) = synth.make_batch(101, B, 1, structured=True, reward_p=0.3)
    before, after, act, rew, term, gt, vm = tup
    stp.forward_backward(before.contiguous().to(DEV), after.contiguous().to(DEV), 1, act.to(DEV), rew.float().to(DEV), term.float().to(DEV))
    torch.cuda.synchronize()
    tr = ref_cpu.Trainer(ref_cpu.default_config(), synth.make_state_dict(7))
    tr.target_net.load_state_dict(synth.make_state_dict(8))
    tr.model.set_train()
    tr.optimizer.zero_grad()
    d = {}
    loss = ref_cpu.process_batch(tr.model, tr.target_net, tr.config, tup, detail=d)
    d["before_values"].retain_grad()
    loss.backward()
    print(f"[{dtype}] loss gpu {stp.loss.item():.8f} cpu {loss.item():.8f}")
    print(f"[{dtype}] q_before relerr %.3e" % relerr(stp.q_before, d["before_values"].detach().reshape(B, 15)))
    # dQ
    from video_dqn_amd.engine import C, _lib
    dq_ref = d["before_values"].grad.reshape(B, 15)
    print(f"[{dtype}] dq absmax ref %.3e" % dq_ref.abs().max().item())
    rows = []
    for name, p in tr.model.named_parameters():
        if p.grad is None:
            continue
        s = net.slots[name]
        g = stp.grads[s.offset:s.offset + s.numel].view(s.shape)
        rows.append((name, relerr(g, p.grad), cos(g, p.grad), (g.double().cpu().norm() / p.grad.double().norm()).item()))
    for r in rows:
        print(f"[{dtype}] {r[0]:42s} relmax {r[1]:.3e} cos {r[2]:.6f} normratio {r[3]:.5f}")


if __name__ == "__main__":
    op_errors()
    step_errors("f32")
    step_errors("bf16")
