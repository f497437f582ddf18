#!/usr/bin/env python
"""Per-tensor gradient agreement of one ARCHITECTURE='basic' TD update with the CPU oracle (all elements)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import ref_cpu  # noqa: E402
from video_dqn_amd import synth  # noqa: E402
from video_dqn_amd.engine import NetEngine, TDStepper  # noqa: E402


def main(F=4, B=3, dtype="f32"):
    dev = "cuda"
    net = NetEngine(3, 5, F, False, dtype, 2 * B)
    net.load_tensors(synth.make_state_dict(7, extra_capacity=False, num_frames=F))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    tnet = NetEngine(3, 5, F, False, dtype, 2 * B)
    tnet.load_tensors(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
    tnet.pack_weights(stp.packed_target)
    (tup, _) = synth.make_batch(400 + 10 * F + 1, B, F, structured=True, reward_p=0.3)
    before, after, act, rew, term, gt, vm = tup
    stp.forward_backward(before.contiguous().to(dev), after.contiguous().to(dev), 1, act.to(dev), rew.float().to(dev), term.float().to(dev))
    torch.cuda.synchronize()
    cfg = ref_cpu.default_config(ARCHITECTURE="basic", PANORAMA=F > 1)
    for prec in (torch.float32, torch.float64):
        tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7, extra_capacity=False, num_frames=F), num_frames=F)
        tr.target_net.load_state_dict(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
        tr.model.to(prec)
        tr.target_net.to(prec)
        t2 = (tup[0].to(prec), tup[1].to(prec)) + tuple(tup[2:])
        tr.model.set_train()
        loss = ref_cpu.process_batch(tr.model, tr.target_net, cfg, t2)
        loss.backward()
        print(f"--- oracle {prec}: loss {loss.item():.8f} engine {stp.loss.item():.8f}")
        rows = []
        for name, p in tr.model.named_parameters():
            if p.grad is None:
                continue
            s = net.slots[name]
            g = stp.grads[s.offset:s.offset + s.numel].view(s.shape).double().cpu()
            r = p.grad.double()
            rows.append((((g - r).abs().max() / r.abs().max()).item(), ((g - r).norm() / r.norm()).item(), name))
        rows.sort(reverse=True)
        for mx, l2, name in rows[:12]:
            print(f"  max {mx:.2e}  l2 {l2:.2e}  {name}")
        if prec == torch.float32:
            ref32 = {n: p.grad.double().clone() for n, p in tr.model.named_parameters() if p.grad is not None}
        else:
            worst = max((((ref32[n] - p.grad).abs().max() / p.grad.abs().max()).item(), n) for n, p in tr.model.named_parameters() if p.grad is not None)
            print("  torch f32 vs torch f64 worst max-err:", worst)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
