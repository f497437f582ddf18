#!/usr/bin/env python
"""Phase timeline of the persistent 64-channel kernel (conv64_kernel, layer1: csrc/igemm.hip; diagnostic build:
VDQN_EXTRA_FLAGS=-DVDQN_STAMP VDQN_LIB_OUT=stamp python -m video_dqn_amd.build; run with VDQN_LIB=stamp).  Thread 0 of every
workgroup stamps s_memtime for its first 16 tiles: arrival at the tile's top | barrier passed (window visible) | next window's eight
LDS-DMA pieces issued | held stores + edge bits done (K loop begins) | K loop done (144 MFMAs per wave) | epilogue done.
Workgroups b and b + 256 share a CU (tools/probes/hwid_probe.hip)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402


def overlap(a0, a1, b0, b1):
    return max(0.0, min(a1, b1) - max(a0, b0))


def main():
    raw = C.CDLL(_lib.LIB_PATH)
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n = 512
    x = torch.randn((n, 56, 56, 64), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((64, 3, 3, 64), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(64, device=dev, generator=g) * 0.1
    res = torch.randn((n, 56, 56, 64), device=dev, generator=g).to(torch.bfloat16)
    buf = torch.zeros((512, 16, 8), dtype=torch.int64, device=dev)
    raw.vdqn_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
    for name, kw in (("forward, bias + ReLU (conv1 of a block)", dict(relu=True)), ("forward, bias + residual + ReLU (conv2 of a block)", dict(relu=True, resid=res)),
                     ("data gradient with mask and residual gradient", dict(mode=1, mask=res, resid=res))):
        args = dict(ho=56, wo=56, co=64, r=3, s=3, stride=1, pad=1, bias=(b if kw.get("mode", 0) == 0 else None), **kw)
        for _ in range(3):
            ops.conv2d(x, w, **args)
        torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv2d(x, w, **args); e1.record()
        torch.cuda.synchronize()
        t = buf.cpu().double()
        us = e0.elapsed_time(e1) * 1e3
        print(f"== {name}: {n} frames, launch {us:.1f} us (stamped build) = {2.0 * n * 56 * 56 * 64 * 576 / us / 1e6:.0f} TFLOP/s")
        mid = t[:, 2:14]
        ph = {"wait at the top (window DMA + barrier)": mid[:, :, 1] - mid[:, :, 0], "issue next window (8 pieces)": mid[:, :, 2] - mid[:, :, 1],
              "held stores + edge bits": mid[:, :, 3] - mid[:, :, 2], "K loop (144 MFMAs per wave)": mid[:, :, 4] - mid[:, :, 3],
              "epilogue": mid[:, :, 5] - mid[:, :, 4], "to the next tile's top": t[:, 3:15, 0] - mid[:, :, 5]}
        tot = 0.0
        for k, v in ph.items():
            print(f"   {k:40s} mean {v.mean().item():8.0f} cycles   (p10 {v.flatten().quantile(0.1).item():6.0f}, p90 {v.flatten().quantile(0.9).item():6.0f})")
            tot += v.mean().item()
        print(f"   {'tile period':40s} mean {tot:8.0f} cycles; 2 x 2304 MFMA cycles per tile pair and SIMD = {2 * 2304 / tot:.2f} of it")
        frac = []
        for blk in range(0, 256, 4):
            a, c = t[blk], t[blk + 256]
            for k in range(2, 14):
                a0, a1 = a[k, 3].item(), a[k, 4].item()
                frac.append(sum(overlap(a0, a1, c[j, 3].item(), c[j, 4].item()) for j in range(16)) / max(a1 - a0, 1.0))
        fr = torch.tensor(frac)
        print(f"   share of a workgroup's K-loop time with its CU neighbour also in the K loop: mean {fr.mean().item():.2f}, p10 {fr.quantile(0.1).item():.2f}, p90 {fr.quantile(0.9).item():.2f}")


if __name__ == "__main__":
    main()
