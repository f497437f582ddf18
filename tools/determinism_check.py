#!/usr/bin/env python
"""Loss trajectory of the first steps with the side stream on/off (same seeds): a race would show up as an early,
large divergence; atomics-order noise only as late, small drift."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import synth  # noqa: E402
from video_dqn_amd.engine import NetEngine, TDStepper  # noqa: E402


def run(dtype, overlap, B=64, steps=8, seed=4):
    dev = "cuda"
    net = NetEngine(3, 5, 1, True, dtype, 2 * B)
    net.lib.vdqn_net_set_overlap(net.handle, int(overlap))
    net.load_tensors(synth.make_state_dict(seed))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    before = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    after = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    act = torch.randint(0, 3, (B,), dtype=torch.int64, device=dev, generator=g)
    rew = (torch.rand((B, 5), device=dev, generator=g) < 0.05).float()
    out = []
    for _ in range(steps):
        loss = stp.step(before, after, 0, act, rew, rew.clone())
        out.append(loss.item())
    gn = stp.grads.double().norm().item()
    return out, gn


if __name__ == "__main__":
    for dtype in ("f32", "bf16"):
        for ov in (0, 1, 0, 1):
            losses, gn = run(dtype, ov)
            print(dtype, "overlap", ov, " ".join(f"{x:.7f}" for x in losses), f"| last grad norm {gn:.6e}")
