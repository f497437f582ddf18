#!/usr/bin/env python
"""Per-layer timing of the stride-2 data gradients of one TD update as the engine issues them: gx = mask * (dgrad3x3/2(g_h) +
dgrad1x1/2(g_o) + resid) with column sums, batch 256 (layer2.0, layer3.0, layer4.0 of ResNet-18).  HIP events around every launch
(libvdqn's launch profiler); the floors beside each: MFMA at 2.5 PFLOP/s, HBM at 5 TB/s of the launch's algorithmic bytes.

    python tools/bench_s2d.py            (VDQN_S2DGRAD_WIN=0 for the generic class-tiled kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

LAYERS = [("layer2.0 64<-128 @56", 64, 128, 56), ("layer3.0 128<-256 @28", 128, 256, 28), ("layer4.0 256<-512 @14", 256, 512, 14)]


def main(n=256, reps=30):
    dev, dt = "cuda", torch.bfloat16
    for name, ci, co, h in LAYERS:
        ho = h // 2
        g_h = torch.randn((n, ho, ho, co), device=dev).to(dt)
        g_o = torch.randn((n, ho, ho, co), device=dev).to(dt)
        wd1 = (torch.randn((ci, 3, 3, co), device=dev) * 0.05).to(dt)
        wd2 = (torch.randn((ci, 1, 1, co), device=dev) * 0.05).to(dt)
        mask = torch.randn((n, h, h, ci), device=dev).to(dt)
        res = torch.randn((n, h, h, ci), device=dev).to(dt)
        kw = dict(ho=h, wo=h, co=ci, r=3, s=3, stride=2, pad=1, mode=1, mask=mask, resid=res, want_colsum=True, wt2=wd2, in2=g_o)
        for _ in range(3):
            ops.conv2d(g_h, wd1, **kw)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(reps):
            ops.conv2d(g_h, wd1, **kw)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        flops = 2.0 * n * ho * ho * co * ci * 10
        nbytes = 2.0 * (2 * n * ho * ho * co + 3 * n * h * h * ci)
        for tag, v in prof.items():
            us = 1e3 * v["ms"] / v["launches"]
            print(f"{name:24s} {tag:30s} {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {nbytes / us / 1e3:7.1f} GB/s   floors: mfma {flops / 2.5e9:5.1f} us  hbm {nbytes / 5e6:5.1f} us")


if __name__ == "__main__":
    main(reps=int(os.environ.get("REPS", "30")))
