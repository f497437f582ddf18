#!/usr/bin/env python
"""Which rounding points explain the distance between the bf16 engine's gradients and the bf16-emulating oracle?
One update at batch B; per gradient tensor the relative L2 distance of the ENGINE to four oracle variants (all with the engine's
ReLU masks): full emulation, emulation without the gradient roundings, without the weight rounding, and the plain fp32 oracle;
plus the distance between the emulation variants themselves (how much the gradient roundings move the oracle's own result)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bf16_emulation as emu  # noqa: E402
from test_gpu_fullsize import _engine_head_masks, _engine_run, _oracle_run  # noqa: E402
from test_gpu_engine import _engine_relu_masks  # noqa: E402
from video_dqn_amd import synth  # noqa: E402


def main(B=8, F=1, seed=101):
    (tup, _) = synth.make_batch(seed, B, F, structured=True, reward_p=0.3)
    net, stp = _engine_run("bf16", B, F, tup, deterministic=True)
    eg = stp.grads.cpu().double()
    masks = _engine_relu_masks(net, stp.acts_online, stp.layout_samples, B * F)
    hmasks = _engine_head_masks(net, stp, B, F)
    variants = {}

    def run(tag, fwd=True, bwd=True, weights=True):
        orig = (emu.q_act, emu.q_fwd, emu.q_bwd, emu.q_weight)
        try:
            emu.q_act = lambda x: emu._Round.apply(x, fwd, bwd)
            emu.q_fwd = lambda x: emu._Round.apply(x, fwd, False)
            emu.q_bwd = lambda x: emu._Round.apply(x, False, bwd)
            if not weights:
                emu.q_weight = lambda w: w
            loss, q, g, _ = _oracle_run(F, tup, torch.float32, wrap=lambda m: emu.EmulatedNet(m, masks, None, graph_first_call_only=True, head_masks=hmasks),
                                        wrap_target=lambda t: emu.EmulatedNet(t))
        finally:
            emu.q_act, emu.q_fwd, emu.q_bwd, emu.q_weight = orig
        variants[tag] = (loss, g)

    # ---- element-wise: the engine's stored activations and stored gradients against the full emulation's (and the plain fp32's) ----
    import numpy as np
    from test_gpu_engine import _act
    rec, grec, rec32, grec32 = {}, {}, {}, {}
    _oracle_run(F, tup, torch.float32, wrap=lambda m: emu.EmulatedNet(m, masks, rec, graph_first_call_only=True, grad_record=grec, head_masks=hmasks),
                wrap_target=lambda t: emu.EmulatedNet(t))
    orig = emu.ROUNDING
    try:
        emu.ROUNDING = False
        _oracle_run(F, tup, torch.float32, wrap=lambda m: emu.EmulatedNet(m, masks, rec32, graph_first_call_only=True, grad_record=grec32, head_masks=hmasks))
    finally:
        emu.ROUNDING = orig
    L, n = stp.layout_samples, B * F

    def l2t(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()

    def eng_act(name, c, sp):
        a = _act(net, stp.acts_online, L, name, (L * F, sp, sp, c))[:n].float().cpu()
        return a.view(B, F, sp, sp, c).permute(1, 0, 4, 2, 3)

    def eng_bwd(name, shape):
        off = net.lib.vdqn_net_bwd_offset(net.handle, B, name.encode())
        assert off >= 0, name
        nb = int(np.prod(shape)) * 2
        return stp.bwd[off:off + nb].view(torch.bfloat16).view(shape).float().cpu()
    print("forward, relative L2 of the engine's stored tensor to: the full emulation | the plain fp32 oracle (engine masks)")
    for name in ["pool"] + [f"{k}{b}" for b in range(8) for k in ("h", "o")] + ["ds2", "ds4", "ds6", "f8"]:
        c, sp = rec[name][0].shape[1], rec[name][0].shape[2]
        e = eng_act(name, c, sp)
        print(f"  {name:6s} {l2t(e, torch.stack(rec[name], 0)):.3e} | {l2t(e, torch.stack(rec32[name], 0)):.3e}")
    for name in ("l0", "l1"):
        e = _act(net, stp.acts_online, L, name, (L, rec[name].shape[1]))[:B].float().cpu()
        print(f"  {name:6s} {l2t(e, rec[name]):.3e} | {l2t(e, rec32[name]):.3e}")
    print(f"  q      {l2t(stp.q_before.cpu(), rec['q']):.3e} | {l2t(stp.q_before.cpu(), rec32['q']):.3e}")
    print("backward (stored gradients), relative L2 of the engine's tensor to: the full emulation | the plain fp32 oracle (engine masks)")
    nq = 15
    dq_e = eng_bwd("dq", (B, 64))[:, :nq]
    print(f"  dq     {l2t(dq_e, grec['dq_f32'][0].reshape(B, nq)):.3e} | {l2t(dq_e, grec32['dq_f32'][0].reshape(B, nq)):.3e}")
    for name, shape in (("g_l1", (B, 256)), ("g_l0", (B, 512))):
        print(f"  {name:6s} {l2t(eng_bwd(name, shape), grec[name][0]):.3e} | {l2t(eng_bwd(name, shape), grec32[name][0]):.3e}")

    def slots(lst):  # hooks fire in reverse slot order
        return torch.stack(lst[::-1], 0)
    e = eng_bwd("g_f8", (n, 5, 5, 64)).view(B, F, 5, 5, 64).permute(1, 0, 4, 2, 3)
    print(f"  g_f8   {l2t(e, slots(grec['g_f8'])):.3e} | {l2t(e, slots(grec32['g_f8'])):.3e}")
    for b in range(7, -1, -1):
        sp, c = 56 >> (b // 2), 64 << (b // 2)
        for k in ("g_o", "g_h"):
            e = eng_bwd(f"{k}{b}", (n, sp, sp, c)).view(B, F, sp, sp, c).permute(1, 0, 4, 2, 3)
            print(f"  {k}{b:<3d} {l2t(e, slots(grec[f'{k}{b}'])):.3e} | {l2t(e, slots(grec32[f'{k}{b}'])):.3e}")
    e = eng_bwd("g_pool", (n, 56, 56, 64)).view(B, F, 56, 56, 64).permute(1, 0, 4, 2, 3)
    print(f"  g_pool {l2t(e, slots(grec['g_pool'])):.3e} | {l2t(e, slots(grec32['g_pool'])):.3e}")
    sys.stdout.flush()

    run("full")
    run("no_bwd_round", bwd=False)
    run("no_fwd_round", fwd=False)
    run("no_weight_round", weights=False)
    run("masks_only", fwd=False, bwd=False, weights=False)
    print(f"B={B} F={F}: engine loss {stp.loss.item():.7f}; " + "; ".join(f"{k} {v[0]:.7f}" for k, v in variants.items()))

    def l2(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
    names = [n for n, s in net.slots.items() if s.kind == 0]
    print(f"{'tensor':38s} " + " ".join(f"{k:>15s}" for k in variants) + "   full~no_bwd  full~masks_only")
    tot = {k: [0.0, 0.0] for k in variants}
    for n in names:
        s = net.slots[n]
        ge = eg[s.offset:s.offset + s.numel].view(s.shape)
        row = []
        for k, (_, g) in variants.items():
            row.append(l2(ge, g[n]))
            tot[k][0] += float((ge - g[n]).pow(2).sum())
            tot[k][1] += float(g[n].pow(2).sum())
        print(f"{n:38s} " + " ".join(f"{v:15.4e}" for v in row) + f"   {l2(variants['full'][1][n], variants['no_bwd_round'][1][n]):.4e}   "
              f"{l2(variants['full'][1][n], variants['masks_only'][1][n]):.4e}")
    print("whole gradient: " + " ".join(f"{k} {(t[0] / t[1]) ** 0.5:.4e}" for k, t in tot.items()))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 101)
