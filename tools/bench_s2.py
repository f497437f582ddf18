#!/usr/bin/env python
"""Per-launch times of the stride-2 BasicBlocks' first convolution and 1x1 downsample (layer2.0 / 3.0 / 4.0 at the online pass's 512
frames and the target pass's 256): the two as separate launches and as ONE launch (sibling fused, win9s.hip win9sp_kernel).
Environment (read once per process by libvdqn): VDQN_S2WIN_PERSIST=-1 = round-4 kernel for the unfused call, 0 = one workgroup per
tile, 1 / 2 = persistent above one / two rounds of resident workgroups."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

BLOCKS = [("layer2.0", 64, 128, 56), ("layer3.0", 128, 256, 28), ("layer4.0", 256, 512, 14)]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    return sum(v["ms"] for v in prof.values()) * 1e3 / reps, "+".join(sorted(prof))


def main(reps=30):
    dev, dt = "cuda", torch.bfloat16
    print(f"VDQN_S2WIN_PERSIST={os.environ.get('VDQN_S2WIN_PERSIST', '(default 1)')}")
    tot = {"sep": 0.0, "fused": 0.0}
    for name, ci, co, hi in BLOCKS:
        ho = hi // 2
        for n in (512, 256):
            x = torch.randn((n, hi, hi, ci), device=dev).to(dt)
            w1 = (torch.randn((co, 3, 3, ci), device=dev) * 0.05).to(dt)
            w2 = (torch.randn((co, 1, 1, ci), device=dev) * 0.1).to(dt)
            b = torch.zeros(co, device=dev)
            kw = dict(ho=ho, wo=ho, co=co, r=3, s=3, stride=2, pad=1, bias=b, relu=True)
            t3, tag3 = timed(lambda: ops.conv2d(x, w1, **kw), reps)
            t1, tag1 = timed(lambda: ops.conv2d(x, w2, ho=ho, wo=ho, co=co, r=1, s=1, stride=2, pad=0, bias=b), reps)
            tf, tagf = timed(lambda: ops.conv2d(x, w1, wt2=w2, bias2=b, co2=co, relu2=False, **kw), reps)
            fl3 = 2.0 * n * ho * ho * co * ci * 9
            fl1 = 2.0 * n * ho * ho * co * ci
            print(f"{name} n={n}: 3x3/2 {t3:7.1f} us ({fl3 / t3 / 1e6:6.0f} TF/s, {tag3}) | 1x1/2 {t1:6.1f} us ({tag1}) | sum {t3 + t1:7.1f} | "
                  f"fused {tf:7.1f} us ({(fl3 + fl1) / tf / 1e6:6.0f} TF/s, {tagf})")
            tot["sep"] += t3 + t1
            tot["fused"] += tf
    print(f"per update (three blocks, 512 + 256 frames): separate {tot['sep'] / 1e3:.4f} ms, fused {tot['fused'] / 1e3:.4f} ms")


if __name__ == "__main__":
    main(reps=int(os.environ.get("REPS", "30")))
