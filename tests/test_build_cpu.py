"""Build-level checks that need no GPU (hipcc cross-compiles gfx950 here)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_window_kernels_have_no_scratch_access_inside_mfma_loops():
    """The window kernels (win9.hip, win9s.hip: the dominant kernel of an update and its stride-2 sibling) sit at 250-256 VGPRs.  A
    spill reload inside a K-step waits behind the step's LDS-DMA pieces (vmcnt retires in order) and costs tens of percent
    (profiles/r05c_bench_win9_balanced_one_kernel.txt): every barrier-to-barrier segment with matrix instructions must be free of
    scratch accesses (tools/check_spills.py compiles the sources to ISA and scans them)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_spills
    for src in ("win9.hip", "win9s.hip", "win9d.hip"):
        assert check_spills.check(src) == [], src
