#!/usr/bin/env python
"""Does replaying one TD update as a hipGraph beat queueing its ~110 launches on the engine's streams?

The update (TDStepper.step: vdqn_net_td_forward, three vdqn_net_backward_stage calls with the early Adam, the optimiser tail) is
captured ONCE on a fixed minibatch (torch.cuda.graph in relaxed mode: the engine's internal side streams join the capture through
the fork / join events they already use) and replayed; the same minibatch is stepped eagerly in alternation on the same box.
Timing only: a replay re-issues the captured launch arguments, so Adam's bias-correction scalars are those of the captured step
(the trajectory is NOT the trainer's; nothing here is a product path).

    python tools/graph_probe.py [--rounds 5] [--steps 100]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    from video_dqn_amd import synth
    from video_dqn_amd.engine import NetEngine, TDStepper

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B = args.batch
    net = NetEngine(3, 5, 1, True, "bf16", 2 * B, device=dev)
    net.load_tensors(synth.make_state_dict(4, extra_capacity=True, num_frames=1))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, target_update_interval=10 ** 9)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    before = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    after = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    act = torch.randint(0, 3, (B,), dtype=torch.int64, device=dev, generator=g)
    rew = (torch.rand((B, 5), device=dev, generator=g) < 0.05).float()
    term = rew.clone()

    def eager():
        return stp.step(before, after, 0, act, rew, term)

    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:  # device ramp
        for _ in range(10):
            eager()
        torch.cuda.synchronize()
        n += 10
    out = {"ramp_steps": n}

    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph, capture_error_mode="relaxed"):
            eager()
        torch.cuda.synchronize()
        out["capture"] = "ok"
    except Exception as e:  # the record of WHY a capture is refused is the result then
        out["capture"] = f"{type(e).__name__}: {str(e)[:600]}"
        print(json.dumps(out))
        return

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / args.steps

    rows = {"streams": [], "graph": []}
    for _ in range(args.rounds):
        rows["streams"].append(round(timed(eager), 4))
        rows["graph"].append(round(timed(graph.replay), 4))
    out["ms_per_update"] = rows
    out["median"] = {k: round(statistics.median(v), 4) for k, v in rows.items()}
    out["loss_after"] = float(stp.loss.item())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
