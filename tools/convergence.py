#!/usr/bin/env python
"""bf16-vs-fp32 convergence evidence for the benchmarked dtype (the reference trains in fp32, train_q_network.py:88-89).

A fixed, LEARNABLE synthetic set (structured frames; the reward of category c is a threshold on the after-frame's mean colour,
so Q-values can actually fit it) is trained for `--steps` updates from identical weights by
  * the HIP engine in bf16 (the benchmarked mode),
  * the HIP engine in f32 (the parity mode),
  * the CPU oracle (oracle/ref_cpu.py, the restated reference) for the first `--oracle-steps` updates,
with the same minibatch index sequence, lr, gamma, clip and target-refresh schedule.  Per-update losses and their EMA
(0.99 / 0.01: the reference's running loss, train_q_network.py:228-231) go to a JSON file; `check()` states the band.

    python tools/convergence.py [--steps 300] [--batch 64] [--out profiles/r02_convergence.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from video_dqn_amd import synth  # noqa: E402


def make_set(n: int, seed: int = 77):
    fb = synth.make_frames_uint8(seed, "before", n, 1, 224, structured=True)
    fa = synth.make_frames_uint8(seed, "after", n, 1, 224, structured=True)
    act = synth.randint(seed, "act", (n,), 3)
    mean_rgb = fa.reshape(n, -1, 3).astype(np.float32).mean(axis=1)  # [n, 3]
    thr = np.array([150.0, 140.0, 130.0, 160.0, 120.0], dtype=np.float32)
    rew = (mean_rgb[:, [0, 1, 2, 0, 1]] > thr[None, :]).astype(np.float32)  # [n, 5], 10-40 % positives
    return fb, fa, act, rew


def ema(xs, a=0.99):
    out, r = [], None
    for x in xs:
        r = x if r is None else r * a + x * (1 - a)
        out.append(r)
    return out


def run_engine(dtype, data, idx_seq, B, lr, gamma, tui, dev="cuda"):
    from video_dqn_amd.engine import NetEngine, TDStepper
    fb, fa, act, rew = data
    net = NetEngine(3, 5, 1, True, dtype, 2 * B, device=dev)
    net.load_tensors(synth.make_state_dict(7))
    stp = TDStepper(net, B, lr=lr, gamma=gamma, clip_rect=True, target_update_interval=tui)
    d_fb, d_fa = torch.from_numpy(fb).to(dev), torch.from_numpy(fa).to(dev)
    d_act, d_rew = torch.from_numpy(act).to(dev), torch.from_numpy(rew).to(dev)
    losses = []
    for idx in idx_seq:
        i = torch.from_numpy(idx).to(dev)
        r = d_rew[i].contiguous()
        loss = stp.step(d_fb[i].contiguous(), d_fa[i].contiguous(), 0, d_act[i].contiguous(), r, r)
        losses.append(loss.item())
    return losses


def run_oracle(data, idx_seq, lr, gamma, tui, threads):
    from oracle import ref_cpu
    torch.set_num_threads(threads)
    fb, fa, act, rew = data
    cfg = ref_cpu.default_config(LEARNING_RATE=lr, GAMMA=gamma, TARGET_UPDATE_INTERVAL=tui)
    tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7))
    losses = []
    for idx in idx_seq:
        r = torch.from_numpy(rew[idx]).long()
        B = len(idx)
        tup = (synth.normalise_frames(fb[idx]), synth.normalise_frames(fa[idx]), torch.from_numpy(act[idx]), r, r.clone(),
               torch.full((B,), float("nan"), dtype=torch.float64), torch.ones((B, 5), dtype=torch.int64))
        losses.append(tr.step(tup))
    return losses


def check(doc, band=0.15):
    """The stated band: at the end of training the bf16 EMA loss is within `band` (relative) of the f32 EMA loss, both
    have fallen to less than half of their first value, and over the oracle's steps the f32 engine tracks the oracle's
    per-update loss within 15 % for the first 12 updates and 30 % afterwards, with a mean signed deviation under 5 %.  (Two fp32 implementations do not stay bit-close under Adam: in its first
    steps every parameter moves by ~lr * sign(g), also those whose gradient is rounding noise, so summation-order differences
    become +-lr differences in a few weights per step; the loss before the first update agrees to 1e-4, later ones to a few
    per cent, without a sign.)"""
    e16, e32 = doc["ema"]["bf16"], doc["ema"]["f32"]
    assert e32[-1] < 0.5 * e32[0] and e16[-1] < 0.5 * e16[0], (e32[0], e32[-1], e16[0], e16[-1])
    rel = abs(e16[-1] - e32[-1]) / e32[-1]
    assert rel <= band, f"bf16 EMA loss {e16[-1]:.5f} vs f32 {e32[-1]:.5f}: {rel:.3f} > {band}"
    # the whole second half of the curve stays inside a (looser) band too: no divergence that happens to cross at the end
    half = len(e32) // 2
    worst = max(abs(a - b) / b for a, b in zip(e16[half:], e32[half:]))
    assert worst <= 2 * band, worst
    lo = doc["loss"].get("oracle") or []
    # the oracle leg: identical before any update; afterwards the two fp32 trajectories drift apart chaotically (see above) but
    # without a bias — the first 12 updates within 15 %, later ones within 30 % (two runs of the SAME engine differ by that
    # much there: its weight-gradient sums are atomic), the mean signed deviation within 5 %
    worst_o, signed = 0.0, []
    for k, (a, b) in enumerate(zip(doc["loss"]["f32"], lo)):
        assert abs(a - b) <= (0.15 if k < 12 else 0.30) * abs(b) + 1e-4, (k, a, b)
        worst_o = max(worst_o, abs(a - b) / abs(b))
        signed.append((a - b) / abs(b))
    if lo:
        assert abs(doc["loss"]["f32"][0] - lo[0]) <= 1e-4 * abs(lo[0]) + 1e-6, (doc["loss"]["f32"][0], lo[0])
        assert abs(sum(signed) / len(signed)) <= 0.05, sum(signed) / len(signed)
    return {"final_rel_diff": rel, "worst_second_half_rel_diff": worst, "oracle_steps_checked": len(lo), "worst_f32_vs_oracle_rel_diff": worst_o,
            "mean_signed_f32_vs_oracle_rel_diff": (sum(signed) / len(signed)) if lo else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--set-size", type=int, default=512)
    ap.add_argument("--oracle-steps", type=int, default=24)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--tui", type=int, default=100)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    t0 = time.time()
    data = make_set(args.set_size)
    idx_seq = [np.sort(synth.randint(1000 + s, "idx", (args.batch,), args.set_size)) for s in range(args.steps)]
    doc = {"what": "per-update TD loss of the same training run in three arithmetic modes (see tools/convergence.py)",
           "config": {"steps": args.steps, "batch": args.batch, "set_size": args.set_size, "lr": args.lr, "gamma": 0.99, "loss_clip": "rect",
                      "target_update_interval": args.tui, "reward_positive_rate": float(data[3].mean())},
           "loss": {}, "ema": {}}
    for dt in ("bf16", "f32"):
        doc["loss"][dt] = run_engine(dt, data, idx_seq, args.batch, args.lr, 0.99, args.tui)
        doc["ema"][dt] = ema(doc["loss"][dt])
    if args.oracle_steps > 0:
        threads = max(1, min(len(os.sched_getaffinity(0)), 32))
        doc["loss"]["oracle"] = run_oracle(data, idx_seq[:args.oracle_steps], args.lr, 0.99, args.tui, threads)
        doc["ema"]["oracle"] = ema(doc["loss"]["oracle"])
    doc["seconds"] = round(time.time() - t0, 1)
    if args.out:  # the curves are the record: written before the band is asserted
        with open(args.out, "w") as f:
            json.dump(doc, f)
    doc["check"] = check(doc)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(doc, f)
    e16, e32 = doc["ema"]["bf16"], doc["ema"]["f32"]
    for k in sorted(set(list(range(0, args.steps, max(1, args.steps // 10))) + [args.steps - 1])):
        o = doc["loss"].get("oracle", [])
        print(f"step {k + 1:4d}  loss bf16 {doc['loss']['bf16'][k]:.5f}  f32 {doc['loss']['f32'][k]:.5f}  "
              f"{'oracle %.5f' % o[k] if k < len(o) else '':16s}  EMA bf16 {e16[k]:.5f}  f32 {e32[k]:.5f}")
    print(json.dumps(doc["check"]))


if __name__ == "__main__":
    main()
