"""Build libvdqn.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python -m video_dqn_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvdqn.so")
SOURCES = ["igemm.hip", "stem.hip", "wgrad.hip", "pointwise.hip", "bn_train.hip", "engine.hip", "profile.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "vdqn.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = os.environ.get("VDQN_EXTRA_FLAGS", "").split()  # e.g. -DVDQN_IGEMM_STAGES=3 for A/B builds
    cmd = [_hipcc()] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
