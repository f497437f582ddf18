#!/usr/bin/env python
"""Per-launch time of the persistent stem kernel (csrc/stem.hip) for the library variant named by VDQN_LIB.  With the diagnostic
builds -DVDQN_STEM_PROBE=1 (no K loop), =2 (no pooling phase), =6 (K loop only), =3 (staging, barriers and the patch phase only) this
separates the kernel's phases: if the full kernel costs about the SUM of the K-loop-only and the epilogue-only builds, the two
workgroups of a CU do not overlap their phases; if it costs about the larger of the two, they do.

    VDQN_LIB=<variant> python tools/stem_phases.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from video_dqn_amd import _lib, ops
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    wt = (torch.randn(64, 256, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device=dev, generator=g) * 0.1
    out = {}
    for name, n, n_idx in (("512 frames, arg-max for 256 (online pass)", 512, 256), ("256 frames, no arg-max (target pass)", 256, 0),
                           ("256 frames, all arg-max", 256, 256)):
        t_in = torch.randn(n, 115, 115, 16, device=dev, generator=g).to(torch.bfloat16)
        pool = torch.empty((n, 56, 56, 64), dtype=torch.bfloat16, device=dev)
        idx = torch.zeros((n, 56, 56, 64), dtype=torch.uint8, device=dev)

        def launch():
            _lib.check(lib.vdqn_stem_conv_pool_n(ops._ptr(t_in), ops._ptr(wt), ops._ptr(bias), ops._ptr(pool), ops._ptr(idx), n, n_idx,
                                                 ops.dtype_code(t_in), ops._stream()), "vdqn_stem_conv_pool_n")
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / reps * 1e3
        del t_in, pool, idx
    print(os.environ.get("VDQN_LIB", "(shipped)"), " | ".join(f"{k}: {v:.1f} us" for k, v in out.items()))


if __name__ == "__main__":
    main()
