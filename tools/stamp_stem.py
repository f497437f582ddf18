#!/usr/bin/env python
"""Phase timeline of the persistent stem kernel (diagnostic build: VDQN_EXTRA_FLAGS=-DVDQN_STAMP VDQN_LIB_OUT=stamp python -m
video_dqn_amd.build; run with VDQN_LIB=stamp).  Thread 0 of every workgroup stamps s_memtime at the phase boundaries of its first 16
tiles: top (window visible) | next window's DMA issued | K loop done | patch written (two barriers) | pooled.  Workgroups b and
b + 256 share a CU (tools/probes/hwid_probe.hip), so their stamps are on one clock: the script reports the mean phase lengths and how
much of a workgroup's K-loop time its neighbour spends in ITS K loop (1.0 = the two run in step, 0.0 = perfectly out of phase)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd import _lib, ops  # noqa: E402


def overlap(a0, a1, b0, b1):
    return max(0.0, min(a1, b1) - max(a0, b0))


def main():
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    wt = (torch.randn(64, 256, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device=dev, generator=g) * 0.1
    n = 256
    t_in = torch.randn(n, 115, 115, 16, device=dev, generator=g).to(torch.bfloat16)
    pool = torch.empty((n, 56, 56, 64), dtype=torch.bfloat16, device=dev)
    idx = torch.zeros((n, 56, 56, 64), dtype=torch.uint8, device=dev)
    buf = torch.zeros((512, 16, 8), dtype=torch.int64, device=dev)
    raw.vdqn_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
    for name, n_idx in (("no arg-max", 0), ("arg-max", n)):
        def launch():
            _lib.check(lib.vdqn_stem_conv_pool_n(ops._ptr(t_in), ops._ptr(wt), ops._ptr(bias), ops._ptr(pool), ops._ptr(idx), n, n_idx,
                                                 ops.dtype_code(t_in), ops._stream()), "vdqn_stem_conv_pool_n")
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        torch.cuda.synchronize()
        b = buf.cpu().double()
        us = e0.elapsed_time(e1) * 1e3
        span = (b[:, :, 4].max() - b[:, 0, 0].min()).item()
        print(f"== {name}: launch {us:.1f} us (stamped build); first top .. last pooled of the stamped tiles {span:.0f} cycles")
        mid = b[:, 2:14]  # tiles 2..13 of every workgroup (steady state)
        ph = {"issue next window": mid[:, :, 1] - mid[:, :, 0], "K loop": mid[:, :, 2] - mid[:, :, 1], "barrier + patch + barrier": mid[:, :, 3] - mid[:, :, 2],
              "pooling": mid[:, :, 4] - mid[:, :, 3], "to next top (wait + barrier)": b[:, 3:15, 0] - mid[:, :, 4]}
        tot = 0.0
        for k, v in ph.items():
            print(f"   {k:30s} mean {v.mean().item():8.0f} cycles   (p10 {v.flatten().quantile(0.1).item():6.0f}, p90 {v.flatten().quantile(0.9).item():6.0f})")
            tot += v.mean().item()
        print(f"   {'tile period':30s} mean {tot:8.0f} cycles")
        # neighbour overlap of the K loops
        frac, slot = [], b[:, 0, 5].long() & 15
        for blk in range(256):
            a, c = b[blk], b[blk + 256]
            for k in range(2, 14):
                a0, a1 = a[k, 1].item(), a[k, 2].item()
                ov = sum(overlap(a0, a1, c[j, 1].item(), c[j, 2].item()) for j in range(16))
                frac.append(ov / max(a1 - a0, 1.0))
        fr = torch.tensor(frac)
        print(f"   share of a workgroup's K-loop time with its CU neighbour also in the K loop: mean {fr.mean().item():.2f}, p10 {fr.quantile(0.1).item():.2f}, p90 {fr.quantile(0.9).item():.2f}"
              f"   (wave slots seen: {sorted(set(slot.tolist()))})")
        for blk in (0, 100):
            print(f"   timeline, blocks {blk} and {blk + 256} (cycles after block {blk}'s tile 2; top / K begin / K end / patch end / pool end):")
            z = b[blk, 2, 0].item()
            for k in range(2, 7):
                ra = " ".join(f"{b[blk, k, i].item() - z:7.0f}" for i in range(5))
                rb = " ".join(f"{b[blk + 256, k, i].item() - z:7.0f}" for i in range(5))
                print(f"      tile {k}:  {ra}   |   {rb}")


if __name__ == "__main__":
    main()
