#!/usr/bin/env python
"""Per-layer timing of the implicit-GEMM kernels on the ResNet-18 shapes of one TD update (batch 256 -> 512 online
images forward, 256 backward): TFLOP/s per layer from HIP events around every launch (libvdqn's launch profiler)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

LAYERS = [  # name, ci, co, hi, k, stride, pad
    ("layer1 3x3 64->64 @56", 64, 64, 56, 3, 1, 1),
    ("layer2 3x3 128->128 @28", 128, 128, 28, 3, 1, 1),
    ("layer2.0 3x3/2 64->128 @56", 64, 128, 56, 3, 2, 1),
    ("layer3 3x3 256->256 @14", 256, 256, 14, 3, 1, 1),
    ("layer4 3x3 512->512 @7", 512, 512, 7, 3, 1, 1),
    ("layer3.0 1x1/2 128->256 @28", 128, 256, 28, 1, 2, 0),
]


def main(n_fwd=512, n_bwd=256, reps=20, dtype=torch.bfloat16):
    dev = "cuda"
    rows = []
    for name, ci, co, hi, k, stride, pad in LAYERS:
        ho = (hi + 2 * pad - k) // stride + 1
        for mode, n in ((0, n_fwd), (1, n_bwd)):
            if mode == 0:
                x = torch.randn((n, hi, hi, ci), device=dev).to(dtype)
                w = (torch.randn((co, k, k, ci), device=dev) * 0.05).to(dtype)
                kw = dict(ho=ho, wo=ho, co=co, r=k, s=k, stride=stride, pad=pad, relu=True, bias=torch.zeros(co, device=dev))
            else:  # data gradient: gy [n, ho, ho, co] -> gx [n, hi, hi, ci]
                x = torch.randn((n, ho, ho, co), device=dev).to(dtype)
                w = (torch.randn((ci, k, k, co), device=dev) * 0.05).to(dtype)
                kw = dict(ho=hi, wo=hi, co=ci, r=k, s=k, stride=stride, pad=pad, mode=1)
            for _ in range(3):
                ops.conv2d(x, w, **kw)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(reps):
                ops.conv2d(x, w, **kw)
            torch.cuda.synchronize()
            prof = _lib.profile_collect()
            _lib.profile_enable(False)
            flops = 2.0 * n * ho * ho * co * ci * k * k
            for tag, v in prof.items():
                us = 1e3 * v["ms"] / v["launches"]
                rows.append((name, "fwd" if mode == 0 else "dgrad", tag, us, flops / us / 1e6))
    for r in rows:
        print(f"{r[0]:30s} {r[1]:6s} {r[2]:28s} {r[3]:8.1f} us  {r[4]:7.1f} TFLOP/s")


if __name__ == "__main__":
    main(n_fwd=int(os.environ.get("N_FWD", "512")), n_bwd=int(os.environ.get("N_BWD", "256")), reps=int(os.environ.get("REPS", "20")))
