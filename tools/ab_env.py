#!/usr/bin/env python
"""Alternating A/B runs of bench.py under different environments on ONE box (variants differ by environment switches or by the
library variant VDQN_LIB): `python tools/ab_env.py [--rounds 3] [--steps 100] name1:K=V,K2=V2 name2: ...` prints tuples/s, ms per
update and the per-kernel ms of every run, then the per-variant medians.  Decisions between variants are taken on these
alternating runs, never on runs from different boxes (the boxes of the pool differ by ~5 %)."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, steps, extra = 3, 100, []
    while args and args[0].startswith("--"):
        if args[0] == "--rounds":
            rounds = int(args[1]); args = args[2:]
        elif args[0] == "--steps":
            steps = int(args[1]); args = args[2:]
        elif args[0] == "--bench-args":
            extra = args[1].split(); args = args[2:]
        else:
            raise SystemExit(f"unknown flag {args[0]}")
    variants = []
    for a in args:
        name, _, envs = a.partition(":")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        variants.append((name, env))
    res = {n: [] for n, _ in variants}
    kern = {n: [] for n, _ in variants}
    for r in range(rounds):
        for name, env in variants:
            e = dict(os.environ, VDQN_BENCH_NO_LIVE_PMC="1")  # (roofline.traffic is not what an A/B compares)
            e.update(env)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "20", "--no-cpu-baseline"] + extra,
                               env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if p.returncode != 0:
                print(name, "FAILED", p.stderr[-1500:])
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[name].append((d["value"], d["ms_per_step"]))
            ks = {k: v["ms_per_step"] for k, v in (d.get("kernels") or {}).items()}
            kern[name].append(ks)
            print(f"round {r} {name:24s} {d['value']:9.1f} tuples/s {d['ms_per_step']:7.3f} ms  dom {d['roofline']['kernel'] if d.get('roofline') else '-'} "
                  f"{d['roofline']['achieved'] if d.get('roofline') else 0:7.1f}", flush=True)
    print("---- medians")
    for name, _ in variants:
        if res[name]:
            print(f"{name:24s} {statistics.median(v for v, _ in res[name]):9.1f} tuples/s {statistics.median(m for _, m in res[name]):7.3f} ms")
    names = sorted({k for n in kern for ks in kern[n] for k in ks}, key=lambda k: -max((statistics.median(ks.get(k, 0) for ks in kern[n]) if kern[n] else 0) for n in kern))
    print("---- per-kernel ms per update (median over rounds; HIP events, side stream serialised)")
    print(f"{'kernel':30s}" + "".join(f"{n:>16s}" for n, _ in variants))
    for k in names:
        print(f"{k:30s}" + "".join(f"{(statistics.median(ks.get(k, 0) for ks in kern[n]) if kern[n] else float('nan')):16.4f}" for n, _ in variants))
    print(f"{'sum':30s}" + "".join(f"{(statistics.median(sum(ks.values()) for ks in kern[n]) if kern[n] else float('nan')):16.4f}" for n, _ in variants))


if __name__ == "__main__":
    main()
