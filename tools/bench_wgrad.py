#!/usr/bin/env python
"""Per-launch times of the window weight-gradient kernel (wgrad_win_kernel: 3x3 / stride 1) on the layer shapes of one TD update at
256 frames, default (f32 atomics) and deterministic (plain stores into split copies + wgrad_reduce): TFLOP/s per layer from HIP events
around every launch.  With VDQN_LIB=<a build with -DVDQN_WGRAD_NO_EMIT> the atomics are skipped (invalid results): the difference to the
default build is what the partial-sum sink costs a launch that has the chip to itself."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

LAYERS = [("layer1 64->64 @56", 64, 56), ("layer2 128->128 @28", 128, 28), ("layer3 256->256 @14", 256, 14), ("layer4 512->512 @7", 512, 7)]


def main(n=256, reps=30):
    dev = "cuda"
    for det in (False, True):
        tot = 0.0
        for name, c, hw in LAYERS:
            x = torch.randn((n, hw, hw, c), device=dev).to(torch.bfloat16)
            gy = torch.randn((n, hw, hw, c), device=dev).to(torch.bfloat16)
            kw = dict(co=c, r=3, s=3, stride=1, pad=1, want_dbias=False, deterministic=det)
            for _ in range(3):
                ops.conv2d_wgrad(gy, x, **kw)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(reps):
                ops.conv2d_wgrad(gy, x, **kw)
            torch.cuda.synchronize()
            prof = _lib.profile_collect()
            _lib.profile_enable(False)
            flops = 2.0 * n * hw * hw * c * c * 9
            parts = ", ".join(f"{k} {1e3 * v['ms'] / reps:6.1f} us" for k, v in sorted(prof.items()))
            us = sum(v["ms"] for v in prof.values()) * 1e3 / reps
            tot += us
            print(f"{'deterministic' if det else 'atomic':13s} {name}: {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s   [{parts}]")
        print(f"{'deterministic' if det else 'atomic':13s} sum over the four layers: {tot:.1f} us")


if __name__ == "__main__":
    main(reps=int(os.environ.get("REPS", "30")))
