"""Shared test helpers (inputs of the golden G4 loss-branch matrix)."""
import numpy as np
import torch

from video_dqn_amd import synth


def g4_inputs(s, Bq=6, A=3):
    qb = torch.from_numpy(synth.uniform(s, "qb", (Bq, 5, A), -1.0, 2.0))
    qo = torch.from_numpy(synth.uniform(s, "qo", (Bq, 5, A), -1.0, 2.0))
    qt = torch.from_numpy(synth.uniform(s, "qt", (Bq, 5, A), -1.0, 2.0))
    qo[0, 0, :] = 1.0
    qo[1, 1, 1:] = 3.0
    act = torch.from_numpy(synth.randint(s, "act", (Bq,), A))
    rew = torch.from_numpy((synth.uniform(s, "rew", (Bq, 5)) < 0.4).astype(np.int64))
    vm = torch.from_numpy((synth.uniform(s, "vm", (Bq, 5)) < 0.7).astype(np.int64))
    return qb, qo, qt, act, rew, rew.clone(), vm


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
