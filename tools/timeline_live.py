#!/usr/bin/env python
"""The update's real timeline: HIP events around every launch of libvdqn with the side stream ON (rocprofv3's kernel trace
serialises the dispatches, so it cannot show this).  Prints, for one steady-state update: wall time, sum of launch spans per stream,
device busy time (union of the spans), idle time, time with both streams busy, the idle gaps by the launch that follows them, and
(--dump) every span.  A span starts when its stream reaches the launch (= the previous launch of that stream ended), so spans of one
stream tile its busy time; the union across streams is what the device was doing.
  python tools/timeline_live.py [--batch 256] [--dump]"""
import argparse
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Span(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("t0", C.c_float), ("t1", C.c_float), ("stream", C.c_int)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dump", action="store_true")
    ap.add_argument("--updates", type=int, default=3)
    ap.add_argument("--force-dist", action="store_true", help="one rank through RCCL: the three gradient buckets are all-reduced (bench.py --force-dist); "
                                                               "VDQN_DIST_EARLY_ADAM=1 then updates stage 0 / 1 behind their own bucket")
    args = ap.parse_args()
    from video_dqn_amd import _lib, synth
    from video_dqn_amd.engine import NetEngine, TDStepper
    dev = torch.device("cuda", 0)
    B = args.batch
    net = NetEngine(3, 5, 1, True, "bf16", 2 * B, device=dev)
    net.load_tensors(synth.make_state_dict(4, extra_capacity=True, num_frames=1))
    comm = None
    if args.force_dist:
        import torch.distributed as dist
        from video_dqn_amd.dist import BucketAllReduce
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        comm = BucketAllReduce(1, force=True)
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, target_update_interval=10 ** 9,
                    allreduce=(comm.launch if comm else None), allreduce_wait=(comm.wait_last if comm else None))
    fin = comm.finish if comm else None
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    b_ = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    a_ = torch.randint(0, 256, (B, 1, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    act = torch.randint(0, 3, (B,), dtype=torch.int64, device=dev, generator=g)
    rew = (torch.rand((B, 5), device=dev, generator=g) < 0.05).float()
    for _ in range(300):
        stp.step(b_, a_, 0, act, rew, rew, finish_allreduce=fin)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(args.updates):
        stp.step(b_, a_, 0, act, rew, rew, finish_allreduce=fin)
    torch.cuda.synchronize()
    raw = C.CDLL(_lib.LIB_PATH)
    buf = (Span * 4096)()
    n = raw.vdqn_debug_profile_timeline(buf, 4096)
    _lib.profile_collect()
    _lib.profile_enable(False)
    spans = [(buf[i].t0 * 1e3, buf[i].t1 * 1e3, buf[i].stream, buf[i].name.decode()) for i in range(n)]
    marks = [i for i, s in enumerate(spans) if s[3].startswith("td_loss")]  # exactly one launch per update
    if len(marks) < 2:
        raise SystemExit("fewer than two updates recorded")
    lo, hi = marks[-2], marks[-1]
    upd = spans[lo:hi]
    t0, t1 = spans[lo][0], spans[hi][0]
    print(f"update (td_loss to the next update's td_loss: backward, Adam, the next forward): {t1 - t0:.1f} us, {len(upd)} launches")
    bys = defaultdict(list)
    for s in upd:
        bys[s[2]].append(s)
    for k, l in sorted(bys.items()):
        print(f"  stream {k}: {len(l)} launches, spans sum {sum(e[1] - e[0] for e in l):.1f} us")
    ev = []
    for s in upd:
        ev.append((max(s[0], t0), 1))
        ev.append((s[1], -1))
    ev.sort()
    depth, last, busy, both = 0, t0, 0.0, 0.0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    print(f"  device busy (union) {busy:.1f} us, idle {t1 - t0 - busy:.1f} us, two or more streams busy {both:.1f} us")
    if args.dump:
        for s in sorted(upd):
            print(f"{s[0] - t0:9.1f} {s[1] - s[0]:8.1f}  s{s[2]}  {s[3]}")


if __name__ == "__main__":
    main()
