#!/usr/bin/env python
"""Per-layer kernel times of one TD update (VDQN_PROFILE_LAYERS=1 rows of the launch profiler), side stream off."""
import os
import sys

os.environ["VDQN_PROFILE_LAYERS"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, synth  # noqa: E402
from video_dqn_amd.engine import NetEngine, TDStepper  # noqa: E402


def main(B=256, F=1, dtype="bf16", steps=5, ec=True):
    dev = "cuda"
    net = NetEngine(3, 5, F, ec, dtype, 2 * B)
    net.load_tensors(synth.make_state_dict(4, extra_capacity=ec, num_frames=F))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    before = torch.randint(0, 256, (B, F, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    after = torch.randint(0, 256, (B, F, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    act = torch.randint(0, 3, (B,), dtype=torch.int64, device=dev, generator=g)
    rew = (torch.rand((B, 5), device=dev, generator=g) < 0.05).float()
    for _ in range(3):
        stp.step(before, after, 0, act, rew, rew)
    net.lib.vdqn_net_set_overlap(net.handle, 0)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(steps):
        stp.step(before, after, 0, act, rew, rew)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    tot = sum(v["ms"] for v in prof.values()) / steps
    print(f"total {tot:.3f} ms/step")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        ms = v["ms"] / steps
        tf = v["flops"] / v["ms"] / 1e9 if v["flops"] > 0 else 0
        print(f"{k:48s} x{v['launches'] // steps:2d} {ms:7.3f} ms  {tf:7.1f} TF  {v['bytes'] / v['ms'] / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main(ec=(len(sys.argv) < 2 or sys.argv[1] != "basic"))
