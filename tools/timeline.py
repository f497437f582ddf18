#!/usr/bin/env python
"""Timeline of one TD update from a rocprofv3 kernel trace (overlap on): how long one / two / more kernels are resident,
and per kernel family the time it spends running alone vs beside another kernel.

    rocprofv3 --kernel-trace -d out -o k --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-profile --no-cpu-baseline
    python tools/timeline.py out/k_kernel_trace.csv
"""
import collections
import csv
import re
import sys


def fam(name):
    m = re.search(r"(\w+_kernel|\w+Buffer\w*|elementwise)", name)
    n = m[1] if m else name[:24]
    m2 = re.search(r"<([^>]*)>", name)
    return n + ("<" + m2[1][:28] + ">" if m2 and "kernel" in n else "")


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) // 2  # which update (default: the middle one: inside bench.py's timed region,
    a, b = adam[k - 1], adam[k]                                   # not its serialised per-kernel profiling leg at the end)
    seg = rows[a + 1:b + 1]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
    ev = []
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, 1, r)); ev.append((e, -1, r))
    ev.sort(key=lambda x: (x[0], x[1]))
    live, last = [], t0
    by_depth = collections.Counter()
    alone, shared = collections.Counter(), collections.Counter()
    for t, d, r in ev:
        dt = t - last
        if dt > 0:
            by_depth[min(len(live), 3)] += dt
            for x in live:
                (alone if len(live) == 1 else shared)[fam(x["Kernel_Name"])] += dt
        last = t
        if d == 1:
            live.append(r)
        else:
            live.remove(r)
    wall = t1 - t0
    print(f"update wall {wall / 1e3:.1f} us, {len(seg)} kernels; resident kernels: 0: {by_depth[0] / 1e3:.0f} us, 1: {by_depth[1] / 1e3:.0f} us, "
          f"2: {by_depth[2] / 1e3:.0f} us, 3+: {by_depth[3] / 1e3:.0f} us")
    print(f"{'kernel':56s} {'alone us':>9s} {'shared us':>10s}")
    for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k])):
        print(f"{k:56s} {alone[k] / 1e3:9.1f} {shared[k] / 1e3:10.1f}")
    # the serial order of the update: first/last launch of the phases
    names = [fam(r["Kernel_Name"]) for r in seg]
    td = next(i for i, n in enumerate(names) if "td_loss" in n)
    print(f"forward (first kernel .. td_loss): {(int(seg[td]['End_Timestamp']) - t0) / 1e3:.1f} us; backward + Adam: {(t1 - int(seg[td]['End_Timestamp'])) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
