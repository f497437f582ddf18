"""Build libvdqn.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python -m video_dqn_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.  Every source is
compiled to its own object (in parallel, only when it or a header changed) and the objects are linked into the
shared library.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
# VDQN_LIB_OUT=<name>: build a variant (A/B or diagnostic builds with VDQN_EXTRA_FLAGS) next to the shipped library
_VARIANT = os.environ.get("VDQN_LIB_OUT", "")
OBJ_DIR = os.path.join(LIB_DIR, "obj", _VARIANT or "main")
LIB_PATH = os.path.join(LIB_DIR, f"libvdqn{'_' + _VARIANT if _VARIANT else ''}.so")
SOURCES = ["igemm.hip", "win9.hip", "win9s.hip", "win9d.hip", "skinny.hip", "stem.hip", "wgrad.hip", "pointwise.hip", "bn_train.hip", "engine.hip", "profile.hip", "comm.hip", "hostio.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "igemm_common.h"), os.path.join(os.path.dirname(HERE), "include", "vdqn.h")]
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    return [h for h in HEADERS if os.path.exists(h)]


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, src.replace(".hip", ".o"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB_PATH, [os.path.join(CSRC, s) for s in _sources()] + _headers())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    extra = os.environ.get("VDQN_EXTRA_FLAGS", "").split()  # e.g. -DVDQN_IGEMM_STAGES=3 for A/B builds
    flag_stamp = os.path.join(OBJ_DIR, ".flags")
    flags_txt = " ".join(CFLAGS + extra)
    if not os.path.exists(flag_stamp) or open(flag_stamp).read() != flags_txt:
        force = True
    hipcc = _hipcc()

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), _obj(src)
        if not force and not _stale(obj, [path] + _headers()):
            return None
        cmd = [hipcc] + CFLAGS + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout)
        return r.stdout

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        for out in ex.map(compile_one, _sources()):
            if out and verbose and out.strip():
                print(out)
    with open(flag_stamp, "w") as f:
        f.write(flags_txt)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in _sources()] + ["-ldl", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
