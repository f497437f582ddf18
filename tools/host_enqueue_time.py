#!/usr/bin/env python
"""Host time of one update's launches vs its GPU time: the host must stay ahead of the device for the three forward chains of an
update to start together (a rocprofv3 trace cannot tell: tracing slows every launch)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import synth  # noqa: E402
from video_dqn_amd.engine import NetEngine, TDStepper  # noqa: E402


def main(B=256, steps=40):
    dev = "cuda"
    net = NetEngine(3, 5, 1, True, "bf16", 2 * B)
    net.load_tensors(synth.make_state_dict(4, extra_capacity=True, num_frames=1))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    before = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    after = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    act = torch.randint(0, 3, (B,), dtype=torch.int64, device=dev, generator=g)
    rew = (torch.rand((B, 5), device=dev, generator=g) < 0.05).float()
    for _ in range(5):
        stp.step(before, after, 0, act, rew, rew)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter()
        stp.step(before, after, 0, act, rew, rew)
        host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.sort()
    print(f"host enqueue per update: median {1e3 * host[len(host) // 2]:.3f} ms, min {1e3 * host[0]:.3f}, max {1e3 * host[-1]:.3f}; "
          f"all {steps} updates enqueued after {1e3 * (t1 - t0):.1f} ms, GPU done after {1e3 * (t2 - t0):.1f} ms ({1e3 * (t2 - t0) / steps:.3f} ms/update)")


if __name__ == "__main__":
    main()
