#!/usr/bin/env python
"""Where a K-step of igemm_win9_kernel spends its time (diagnostic build: VDQN_EXTRA_FLAGS=-DVDQN_STAMP VDQN_LIB_OUT=stamp
python -m video_dqn_amd.build; run with VDQN_LIB=stamp).  Wave 0 of every workgroup sums s_memtime deltas over its K loop:
wait = s_waitcnt vmcnt(0) lgkmcnt(0) at the top of a step (DMA of the next tile + own fragment reads), barrier = s_barrier,
issue = the step's LDS-DMA instructions, compute = fragment reads + MFMAs (issue side).  Cycles are shader clocks."""
import ctypes as C
import os
import sys

os.environ.setdefault("VDQN_WIN9_PERSIST", "0")  # the per-K-step averages assume one tile per workgroup

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

LAYERS = [("layer2 128->128 @28", 128, 128, 28), ("layer3 256->256 @14", 256, 256, 14), ("layer4 512->512 @7", 512, 512, 7)]


def main(n=512):
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dev = "cuda"
    for name, ci, co, hw in LAYERS:
        x = torch.randn((n, hw, hw, ci), device=dev).to(torch.bfloat16)
        w = (torch.randn((co, 3, 3, ci), device=dev) * 0.05).to(torch.bfloat16)
        kw = dict(ho=hw, wo=hw, co=co, r=3, s=3, stride=1, pad=1, relu=True)
        grid = ((n * hw * hw + 127) // 128) * (co // 128)
        buf = torch.zeros((grid, 16), dtype=torch.int64, device=dev)
        raw.vdqn_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
        for _ in range(3):
            ops.conv2d(x, w, **kw)
        torch.cuda.synchronize()
        buf.zero_()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        ops.conv2d(x, w, **kw)
        t1.record()
        torch.cuda.synchronize()
        b = buf.cpu().double()
        nk = b[:, 7].mean().item()
        total = (b[:, 2] - b[:, 0])
        loop = (b[:, 1] - b[:, 0])
        epi = (b[:, 2] - b[:, 1])
        span = (b[:, 2].max() - b[:, 0].min()).item()
        flops = 2.0 * n * hw * hw * co * ci * 9
        us = t0.elapsed_time(t1) * 1e3
        print(f"{name}: {grid} workgroups x {nk:.0f} K-steps, launch {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s (stamped build), "
              f"first-begin..last-end {span:.0f} cycles = {span / us / 1e3:.2f} GHz if one clock domain")
        print(f"   per workgroup (cycles): total {total.mean():.0f}  prologue+loop {loop.mean():.0f}  epilogue {epi.mean():.0f}")
        cols = (("wait", 3), ("barrier", 4), ("issue", 5), ("compute", 6))
        for lbl, col in cols:
            print(f"   per K-step {lbl:12s} {b[:, col].mean().item() / nk:8.1f}   (min over workgroups {b[:, col].min().item() / nk:7.1f}, max {b[:, col].max().item() / nk:7.1f})")
        ghz = ((b[:, 2] - b[:, 0]) / ((b[:, 9] - b[:, 8]).clamp_min(1.0) * 10.0))  # cycles per ns: realtime ticks are 10 ns
        print(f"   in-kernel shader clock (s_memtime / s_memrealtime): median {ghz.median().item():.3f} GHz, min {ghz.min().item():.3f}, max {ghz.max().item():.3f}")
        start = b[:, 0] - b[:, 0].min()
        order = torch.argsort(start)
        q = [start[order[int(f * (grid - 1))]].item() for f in (0.0, 0.25, 0.5, 0.75, 1.0)]
        print(f"   workgroup start times (cycles after the first): quartiles {q}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 512)
