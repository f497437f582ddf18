#!/usr/bin/env python
"""A/B of update schedules INSIDE one process: the modes alternate window by window on one engine, so the box's slow drift
(profiles/r6_03_mode_probe_run_to_run_spread.txt: 5.42-5.70 ms from process to process, and inside one) hits them alike.  Only for
switches that are read per call (module attributes of video_dqn_amd.engine, the `next_frames` announcement):
    python tools/ab_inproc.py [--rounds 8] [--steps 40] base ahead pack separate ...
modes:  base      default engine, no announcement of the next minibatch
        pack      next minibatch announced and packed under this update's backward pass (TDStepper.step(next_frames=...))
        <name>:<attr>=<value>[,<attr>=<value>]   base with module attributes of video_dqn_amd.engine set, e.g. late:_EARLY_ADAM=False;
                  lib.<setter>=<int> calls a debug setter of libvdqn instead, e.g. bm128:lib.vdqn_debug_set_win9_bm256=0 (reset to -1 afterwards)
(Round 6 used it with two more modes — a fused Adam + weight-fold kernel and the next update's target pass run ahead — both measured
slower: experiments/r6_fused_adam_fold_and_target_ahead.patch, profiles/r6_05_ab_inproc_*.txt.)"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("modes", nargs="+")
    args = ap.parse_args()
    from video_dqn_amd import engine as eng, synth
    dev = torch.device("cuda", 0)
    B = args.batch
    net = eng.NetEngine(3, 5, 1, True, "bf16", 2 * B, device=dev)
    net.load_tensors(synth.make_state_dict(4, extra_capacity=True, num_frames=1))
    stp = eng.TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, target_update_interval=1000000)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    pool = []
    for _ in range(4):
        b_ = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
        a_ = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
        act_ = torch.randint(0, 3, (B,), dtype=torch.int64, device=dev, generator=g)
        rew_ = (torch.rand((B, 5), device=dev, generator=g) < 0.05).float()
        pool.append((b_, a_, act_, rew_, rew_.clone()))
    k = {"i": 0}

    def step(announce):
        b_, a_, act_, rew_, term_ = pool[k["i"] % 4]
        k["i"] += 1
        nxt = (pool[k["i"] % 4][0], pool[k["i"] % 4][1], 0, True) if announce else None
        stp.step(b_, a_, 0, act_, rew_, term_, next_frames=nxt)

    defaults, lib_set = {}, set()
    import ctypes as C
    from video_dqn_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)

    def setup(mode):
        for k_, v_ in defaults.items():
            setattr(eng, k_, v_)
        for fn in lib_set:
            getattr(raw, fn)(C.c_int(-1))
        name, _, attrs = mode.partition(":")
        for kv in filter(None, attrs.split(",")):
            k_, v_ = kv.split("=", 1)
            if k_.startswith("lib."):
                lib_set.add(k_[4:])
                getattr(raw, k_[4:])(C.c_int(int(v_)))
                continue
            defaults.setdefault(k_, getattr(eng, k_))
            setattr(eng, k_, {"True": True, "False": False}.get(v_, v_))
        return name == "pack"
    for _ in range(150):  # ramp
        step(False)
    torch.cuda.synchronize()
    res = {m: [] for m in args.modes}
    for r in range(args.rounds):
        for m in (args.modes if r % 2 == 0 else args.modes[::-1]):
            ann = setup(m)
            for _ in range(8):
                step(ann)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(ann)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            res[m].append(dt)
            print(f"round {r} {m:16s} {dt:7.3f} ms  {B / dt * 1e3:9.1f} tuples/s", flush=True)
    print("---- medians (ms per update, tuples/s)")
    for m in args.modes:
        med = statistics.median(res[m])
        print(f"{m:16s} {med:7.3f} ms  {B / med * 1e3:9.1f}   min {min(res[m]):.3f} max {max(res[m]):.3f}")


if __name__ == "__main__":
    main()
