#!/usr/bin/env python
"""Per-launch times of the dominant kernel (win9u_kernel: 3x3 / stride 1, 128+ channels) on the shapes and frame counts of one
TD update — forward at 512 (online pass) and 256 (target pass) frames, 384 / 192 (12-view config at batch 16): TFLOP/s per layer from
HIP events around every launch (libvdqn's launch profiler).  Environment (read once per process): VDQN_WIN9_BALANCED=0 = the
static tile walk of rounds 2-4, 1 (default) = rows split evenly over the resident workgroups for launches of more than one round."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

LAYERS = [("layer2 128->128 @28", 128, 28), ("layer3 256->256 @14", 256, 14), ("layer4 512->512 @7", 512, 7)]


def main(reps=30):
    dev, dt = "cuda", torch.bfloat16
    print(f"VDQN_WIN9_BALANCED={os.environ.get('VDQN_WIN9_BALANCED', '(default 1)')}")
    frames = [int(f) for f in os.environ.get("FRAMES", "512,256,384,192").split(",")]
    tot = {f: 0.0 for f in frames}
    for name, c, hw in LAYERS:
        for n in frames:
            x = torch.randn((n, hw, hw, c), device=dev).to(dt)
            w = (torch.randn((c, 3, 3, c), device=dev) * 0.05).to(dt)
            res = torch.randn((n, hw, hw, c), device=dev).to(dt)
            kw = dict(ho=hw, wo=hw, co=c, r=3, s=3, stride=1, pad=1, relu=True, resid=res, bias=torch.zeros(c, device=dev))
            for _ in range(3):
                ops.conv2d(x, w, **kw)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(reps):
                ops.conv2d(x, w, **kw)
            torch.cuda.synchronize()
            prof = _lib.profile_collect()
            _lib.profile_enable(False)
            us = sum(v["ms"] for v in prof.values()) * 1e3 / reps
            flops = 2.0 * n * hw * hw * c * c * 9
            tiles = ((n * hw * hw + 127) // 128) * (c // 128)
            print(f"{name} n={n}: {tiles} tiles = {tiles / 512:.2f} rounds, {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")
            tot[n] += us
    print("sum over the three layers (us): " + ", ".join(f"n={n}: {t:.1f}" for n, t in tot.items()))


if __name__ == "__main__":
    main(reps=int(os.environ.get("REPS", "30")))
