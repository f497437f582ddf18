#!/usr/bin/env python
"""Where a tile of win9sp_kernel (stride-2 plane-window kernel, persistent, downsample fused) spends its time.  Diagnostic build:
    VDQN_EXTRA_FLAGS=-DVDQN_STAMP VDQN_LIB_OUT=stamp python -m video_dqn_amd.build;   run with VDQN_LIB=stamp.
Wave 0 of every workgroup sums s_memtime deltas over its tiles: the wait at the top of a K-step (split: first step behind an
epilogue, second step, all others), the barrier, the DMA issue, fragment reads + MFMAs (issue side), the phase boundary's wait +
barrier, the two epilogues, the tile's top.  Cycles are shader clocks; one MFMA-only K-step is 512 cycles per wave."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

BLOCKS = [("layer2.0", 64, 128, 56), ("layer3.0", 128, 256, 28), ("layer4.0", 256, 512, 14)]


def main():
    _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dev, dt = "cuda", torch.bfloat16
    for name, ci, co, hi in BLOCKS:
        ho = hi // 2
        for n in (512, 256):
            for sib in (True, False):
                if not sib and ci == 64:
                    continue  # (one chunk without the sibling: the round-4 kernel, no stamps)
                x = torch.randn((n, hi, hi, ci), device=dev).to(dt)
                w1 = (torch.randn((co, 3, 3, ci), device=dev) * 0.05).to(dt)
                w2 = (torch.randn((co, 1, 1, ci), device=dev) * 0.1).to(dt)
                b = torch.zeros(co, device=dev)
                kw = dict(ho=ho, wo=ho, co=co, r=3, s=3, stride=2, pad=1, bias=b, relu=True)
                if sib:
                    kw.update(wt2=w2, bias2=b, co2=co, relu2=False)
                buf = torch.zeros((8192, 16), dtype=torch.int64, device=dev)
                raw.vdqn_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
                for _ in range(3):
                    ops.conv2d(x, w1, **kw)
                torch.cuda.synchronize()
                buf.zero_()
                t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
                t0.record()
                ops.conv2d(x, w1, **kw)
                t1.record()
                torch.cuda.synchronize()
                bb = buf.cpu().double()
                bb = bb[bb[:, 2] > 0]
                tiles = bb[:, 2]
                us = t0.elapsed_time(t1) * 1e3
                flops = 2.0 * n * ho * ho * co * ci * (10 if sib else 9)
                steps = (ci // 64) * (10 if sib else 9)
                total = bb[:, 1] - bb[:, 0]
                ghz = (total / ((bb[:, 11] - bb[:, 10]).clamp_min(1.0) * 10.0)).median().item()
                print(f"{name} n={n} {'fused' if sib else 'plain'}: {len(bb)} workgroups, {tiles.sum():.0f} tiles ({tiles.min():.0f}-{tiles.max():.0f} each), "
                      f"{steps} K-steps per tile, launch {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s (stamped build), clock {ghz:.2f} GHz")
                per_tile = total.sum() / tiles.sum()
                cols = (("top", 12), ("wait(first step)", 13), ("wait(second step)", 14), ("wait(other steps)", 3), ("barrier", 4), ("issue", 5),
                        ("reads+MFMA", 6), ("boundary", 9), ("epilogue 3x3", 7), ("epilogue 1x1", 8))
                line = "   cycles per tile: total %.0f = " % per_tile.item()
                line += ", ".join(f"{lbl} {bb[:, c].sum().item() / tiles.sum().item():.0f}" for lbl, c in cols)
                print(line + f"; MFMA-only would be {steps * 512}")


if __name__ == "__main__":
    main()
