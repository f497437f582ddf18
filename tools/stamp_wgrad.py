#!/usr/bin/env python
"""Where a K tile (128 pixels) of wgrad_win_kernel spends its time (diagnostic build: VDQN_EXTRA_FLAGS=-DVDQN_STAMP VDQN_LIB_OUT=stamp
python -m video_dqn_amd.build; run with VDQN_LIB=stamp).  Thread 0 of every workgroup sums s_memtime deltas over its K loop:
wait = s_waitcnt vmcnt(0) at the top of a tile (the LDS-DMA of this tile, issued one tile earlier), barrier = s_barrier, issue = the
next tile's LDS-DMA instructions, compute = fragment reads + MFMAs of the tile (issue side).  Cycles are shader clocks."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402

LAYERS = [("layer1 64->64 @56", 64, 64, 56), ("layer2 128->128 @28", 128, 128, 28), ("layer3 256->256 @14", 256, 256, 14), ("layer4 512->512 @7", 512, 512, 7)]


# (round 5) the launches that stay on the generic wgrad_kernel: the 3x3 / stride-2 convolutions and the 1x1 / stride-2 downsamples
GENERIC = [("layer2.0.conv1 3x3/2 64->128 @56", 64, 128, 56, 3), ("layer3.0.conv1 3x3/2 128->256 @28", 128, 256, 28, 3),
           ("layer4.0.conv1 3x3/2 256->512 @14", 256, 512, 14, 3), ("layer3.0.downsample 1x1/2 128->256 @28", 128, 256, 28, 1),
           ("layer4.0.downsample 1x1/2 256->512 @14", 256, 512, 14, 1)]


def main(n=256, generic=False):
    _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dev = "cuda"
    for row in (GENERIC if generic else LAYERS):
        name, ci, co, hw = row[:4]
        k = row[4] if generic else 3
        ho = hw // 2 if generic else hw
        x = torch.randn((n, hw, hw, ci), device=dev).to(torch.bfloat16)
        gy = torch.randn((n, ho, ho, co), device=dev).to(torch.bfloat16)
        kw = dict(co=co, r=k, s=k, stride=2 if generic else 1, pad=(k // 2 if generic else 1), want_dbias=False)
        buf = torch.zeros((4096, 16), dtype=torch.int64, device=dev)
        raw.vdqn_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
        for _ in range(3):
            ops.conv2d_wgrad(gy, x, **kw)
        torch.cuda.synchronize()
        buf.zero_()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        ops.conv2d_wgrad(gy, x, **kw)
        t1.record()
        torch.cuda.synchronize()
        b = buf.cpu().double()
        b = b[b[:, 7] > 0]
        grid = b.shape[0]
        nk = b[:, 7].mean().item()
        us = t0.elapsed_time(t1) * 1e3  # (includes the zero-fill of dw)
        flops = 2.0 * n * ho * ho * co * ci * k * k
        total, loop, epi = b[:, 2] - b[:, 0], b[:, 1] - b[:, 0], b[:, 2] - b[:, 1]
        print(f"{name}: {grid} workgroups x {nk:.1f} K tiles, call {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s (stamped build, dw zero-fill included)")
        print(f"   per workgroup (cycles): total {total.mean():.0f}  prologue+loop {loop.mean():.0f}  epilogue (LDS sum + atomics) {epi.mean():.0f}")
        for lbl, col in (("wait", 3), ("barrier", 4), ("issue", 5), ("compute", 6)):
            print(f"   per K tile {lbl:10s} {b[:, col].mean().item() / nk:8.1f}   (min over workgroups {b[:, col].min().item() / nk:7.1f}, max {b[:, col].max().item() / nk:7.1f})")
        ghz = (b[:, 2] - b[:, 0]) / ((b[:, 9] - b[:, 8]).clamp_min(1.0) * 10.0)
        print(f"   in-kernel shader clock (s_memtime / s_memrealtime): median {ghz.median().item():.3f} GHz")


def stem(n=256):
    """stem_wgrad_pool_kernel: conv1's weight gradient from the pooled gradient, one step = one pooled row (two conv rows)."""
    _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dev = "cuda"
    g_pool = torch.randn((n, 56, 56, 64), device=dev).to(torch.bfloat16)
    idx = torch.randint(0, 9, (n, 56, 56, 64), dtype=torch.uint8, device=dev)
    t_in = torch.randn((n, 115, 115, 16), device=dev).to(torch.bfloat16)
    buf = torch.zeros((4096, 16), dtype=torch.int64, device=dev)
    raw.vdqn_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
    for _ in range(3):
        ops.stem_wgrad_pool(g_pool, idx, t_in)
    torch.cuda.synchronize()
    buf.zero_()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    ops.stem_wgrad_pool(g_pool, idx, t_in)
    t1.record()
    torch.cuda.synchronize()
    b = buf.cpu().double()
    b = b[b[:, 7] > 0]
    nk = b[:, 7].mean().item()
    us = t0.elapsed_time(t1) * 1e3
    print(f"wgrad_stem_pool n={n}: {b.shape[0]} workgroups x {nk:.1f} steps (pooled rows), call {us:.1f} us (stamped build, dw zero-fill included) = {2.0 * n * 112 * 112 * 64 * 147 / us / 1e6:.0f} TFLOP/s")
    print(f"   per workgroup (cycles): start .. loop end {(b[:, 1] - b[:, 0]).mean():.0f}")
    for lbl, col in (("wait + barrier [A]", 3), ("x DMA issue + tile build", 5), ("barrier [B]", 4), ("pool DMA issue + MFMAs", 6)):
        print(f"   per step {lbl:26s} {b[:, col].mean().item() / nk:8.1f}   (min over workgroups {b[:, col].min().item() / nk:7.1f}, max {b[:, col].max().item() / nk:7.1f})")
    print(f"   MFMA-only per step and wave: 128 x 16 = 2048 cycles")
    ghz = (b[:, 2] - b[:, 0]) / ((b[:, 9] - b[:, 8]).clamp_min(1.0) * 10.0)
    print(f"   in-kernel shader clock: median {ghz.median().item():.3f} GHz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stem":
        stem()
    else:
        main(256, generic=len(sys.argv) > 1 and sys.argv[1] == "generic")
