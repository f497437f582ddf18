"""bf16 trains like fp32 (tools/convergence.py): a short version of the committed 300-update record
(profiles/r02_convergence.json) — identical start, identical minibatch sequence; the bf16 loss curve must stay inside the
stated band of the f32 curve and the f32 engine must track the CPU oracle update by update."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_loss_curve_tracks_f32_and_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import convergence as cv
    from video_dqn_amd import synth
    steps, B, n = 120, 32, 256
    data = cv.make_set(n)
    idx_seq = [np.sort(synth.randint(1000 + s, "idx", (B,), n)) for s in range(steps)]
    doc = {"loss": {}, "ema": {}}
    for dt in ("bf16", "f32"):
        doc["loss"][dt] = cv.run_engine(dt, data, idx_seq, B, 1e-4, 0.99, 50)
        doc["ema"][dt] = cv.ema(doc["loss"][dt])
    doc["loss"]["oracle"] = cv.run_oracle(data, idx_seq[:6], 1e-4, 0.99, 50, max(1, min(len(os.sched_getaffinity(0)), 32)))
    res = cv.check(doc)
    import warnings
    warnings.warn(f"convergence (120 updates, batch 32): EMA loss bf16 {doc['ema']['bf16'][-1]:.5f} vs f32 {doc['ema']['f32'][-1]:.5f} "
                  f"(first {doc['ema']['f32'][0]:.5f}); {res}")
