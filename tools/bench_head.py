#!/usr/bin/env python
"""Per-launch time of the Q-head's layers (archs/HabitatDQNMultiAction.py:30-31) through vdqn_conv2d at the update's shapes:
features.8 at 512 / 256 frames, top.0 / top.2 / top.4 forward at 512 samples, their data gradients at 256.  HIP events around
`reps` back-to-back launches.  Kernel selection by environment (VDQN_SKINNY, VDQN_SKINNY_CONV_CFG): run once per variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    tag = f"SKINNY={os.environ.get('VDQN_SKINNY', '1')} CONV_CFG={os.environ.get('VDQN_SKINNY_CONV_CFG', '0')}"
    out = []
    for n in (512, 256):
        x = torch.randn(n, 7, 7, 512, device=DEV).to(BF)
        w = (torch.randn(64, 3, 3, 512, device=DEV) * 0.02).to(BF)
        b = torch.zeros(64, device=DEV)
        t = timeit(lambda: ops.conv2d(x, w, ho=5, wo=5, co=64, r=3, s=3, stride=1, pad=0, bias=b, relu=True))
        out.append(f"f8@{n} {t:6.1f} us ({2.0 * n * 25 * 64 * 4608 / t / 1e6:6.1f} TF/s)")
    for name, B, fin, fout in (("top0", 512, 1600, 512), ("top2", 512, 512, 256), ("top4", 512, 256, 64)):
        x = torch.randn(B, 1, 1, fin, device=DEV).to(BF)
        w = (torch.randn(fout, 1, 1, fin, device=DEV) * 0.02).to(BF)
        b = torch.zeros(fout, device=DEV)
        t = timeit(lambda: ops.conv2d(x, w, ho=1, wo=1, co=fout, r=1, s=1, stride=1, pad=0, bias=b, relu=True))
        out.append(f"{name}@{B} {t:6.1f} us")
    for name, B, fin, fout in (("top4_dgrad", 256, 64, 256), ("top2_dgrad", 256, 256, 512), ("top0_dgrad", 256, 512, 1600)):
        gy = torch.randn(B, 1, 1, fin, device=DEV).to(BF)
        wd = (torch.randn(fout, 1, 1, fin, device=DEV) * 0.02).to(BF)
        act = torch.randn(B, 1, 1, fout, device=DEV).clamp_min(0).to(BF)
        t = timeit(lambda: ops.conv2d(gy, wd, ho=1, wo=1, co=fout, r=1, s=1, stride=1, pad=0, mode=1, mask=act, want_colsum=True))
        out.append(f"{name}@{B} {t:6.1f} us")
    print(tag + ": " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
