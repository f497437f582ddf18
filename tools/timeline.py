#!/usr/bin/env python
"""Timeline of one steady-state update from a rocprofv3 --kernel-trace CSV (bench.py, side stream on): per queue the busy time,
the union of all kernel intervals (GPU busy), the idle time, and the gaps on the main queue by the kernel that follows them.
  python tools/timeline.py <k_kernel_trace.csv> [update_index_from_end=3]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:52]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])) for r in rows]
    ev.sort()
    adam = [i for i, e in enumerate(ev) if e[3].startswith("adam_kernel")]
    a0, a1 = adam[-back - 1], adam[-back]
    upd = ev[a0 + 1:a1 + 1]
    t0, t1 = ev[a0][1], ev[a1][1]
    print(f"update: {len(upd)} dispatches, {(t1 - t0) / 1e3:.1f} us from the end of one adam to the end of the next")
    byq = defaultdict(list)
    for e in upd:
        byq[e[2]].append(e)
    for qid, l in byq.items():
        print(f"  queue {qid}: {len(l)} dispatches, busy {sum(e[1] - e[0] for e in l) / 1e3:.1f} us")
    # union
    cur_s, cur_e, busy = None, None, 0
    for s, e, _, _ in sorted(upd):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e[1] - e[0] for e in upd)
    print(f"  sum of kernel durations {tot / 1e3:.1f} us; GPU busy (union) {busy / 1e3:.1f} us; idle {(t1 - t0 - busy) / 1e3:.1f} us; "
          f"overlapped {(tot - busy) / 1e3:.1f} us")
    # gaps on the busiest queue
    mainq = max(byq, key=lambda k: len(byq[k]))
    l = sorted(byq[mainq])
    gaps = defaultdict(lambda: [0, 0.0])
    prev_end = t0
    for s, e, _, n in l:
        g = s - prev_end
        if g > 0:
            gaps[n][0] += 1
            gaps[n][1] += g
        prev_end = max(prev_end, e)
    print(f"  gaps on queue {mainq} before a kernel (count, total us):")
    for n, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {n:54s} {c:3d} {g / 1e3:8.1f}")
    print(f"    total gap {sum(g for _, g in gaps.values()) / 1e3:.1f} us")
    if len(sys.argv) > 3:
        for s, e, qd, n in sorted(upd):
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{qd}  {n}")


if __name__ == "__main__":
    main()
