#!/usr/bin/env python
"""No scratch (spill) access inside an MFMA loop.  A scratch reload in a K-step of the window kernels returns only behind the
LDS-DMA pieces issued before it (vmcnt retires in order), i.e. it exposes the whole staging latency: the one-kernel balanced walk
of round 5 lost 60 % to five such reloads per 18 steps (profiles/r05c_bench_win9_balanced_one_kernel.txt).  The kernels sit at
256 VGPRs, so any edit can tip the allocator: this check compiles the given sources to ISA and fails if a barrier-to-barrier
segment with matrix instructions touches scratch.

    python tools/check_spills.py [win9.hip win9s.hip ...]        (default: the window kernels)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video_dqn_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S", "--cuda-device-only"]


def check(src, min_mfma=8):
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([hipcc] + FLAGS + [os.path.join(CSRC, src), "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        txt = open(out).read()
    bad = []
    for fn in re.split(r"\n(?=_Z\w+:)", txt):
        name = fn.split(":")[0]
        if not name.startswith("_Z"):
            continue
        seg, k = [], 0
        for line in fn.split("\n"):
            seg.append(line)
            if "s_barrier" in line:
                mf = sum("v_mfma" in l for l in seg)
                # (accesses behind the segment's last matrix instruction are loop-exit code: the segment of a loop's final step
                # runs on to the first barrier behind the loop)
                last = max((i for i, l in enumerate(seg) if "v_mfma" in l), default=-1)
                sc = sum("scratch_" in l for l in seg[:last + 1])
                if mf >= min_mfma and sc:
                    bad.append((name, k, mf, sc))
                seg, k = [], k + 1
    return bad


def main():
    srcs = sys.argv[1:] or ["win9.hip", "win9s.hip", "win9d.hip"]
    rc = 0
    for s in srcs:
        bad = check(s)
        for name, k, mf, sc in bad:
            print(f"{s}: {name}: segment {k} has {mf} MFMAs and {sc} scratch accesses")
            rc = 1
        if not bad:
            print(f"{s}: no scratch access inside MFMA segments")
    return rc


if __name__ == "__main__":
    sys.exit(main())
