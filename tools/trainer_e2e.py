#!/usr/bin/env python
"""End-to-end throughput of the actual trainer loop (train_q_network.py:211-234 as `video_dqn_amd.trainer.run_train` runs it:
minibatch gather / loader, TDStepper.step, asynchronous loss ring, progress print) next to `bench.py` on the same box in the same
job.  Writes one JSON record:

    python tools/trainer_e2e.py --out profiles/r04_trainer_e2e.json [--steps 2000] [--loader-steps 300]

  resident : `python train_q_network.py <cfg>` with BATCH_SIZE 256, decoded-frame shards, DEVICE_RESIDENT_DATA on — tuples/s from the
             wall clock of the steps after a warm-up run of the same command (the process start, model build and shard upload are
             timed separately and reported, not hidden);
  loader   : the same with DEVICE_RESIDENT_DATA off: the STREAMING path (round 5: memory-mapped shards, native gather into pinned
             staging buffers, H2D on a prefetch stream — shards.HostFrameStream);
  dataloader: SHARD_INPUT 'dataloader' with NUM_WORKERS workers (rounds 1-4: worker -> shared memory -> pinned memory -> H2D), for contrast;
  bench    : `python bench.py --steps 100 --warmup 20` (resident 4-minibatch pool).
The dataset is synthetic (random uint8 frames: nothing is decoded on either path, so the loader figure is an upper bound for
real JPEG data)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_cfg(folder, shards, steps, resident, workers, batch, shard_input="stream"):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "config.yml"), "w") as f:
        f.write(f"DATASET: '{shards}'\nPANORAMA: False\nLOSS_CLIP: 'rect'\nARCHITECTURE: 'extra_capacity'\nLEARNING_RATE: 0.0001\nGAMMA: 0.99\n"
                f"USE_INVERSE_ACTIONS: True\nCHECKPOINT_INTERVAL: 100000000\nNUM_STEPS: {steps}\nSEED: 4\nBATCH_SIZE: {batch}\n"
                f"NUM_WORKERS: {workers}\nCOMPUTE_DTYPE: 'bf16'\nTARGET_UPDATE_INTERVAL: 1000\nDEVICE_RESIDENT_DATA: '{'on' if resident else 'off'}'\n"
                f"SHARD_INPUT: '{shard_input}'\n")


def timed_run(folder):
    """Run the CLI; the trainer prints `batch:<n>/<N> avg_loss: ...` with \\r every update — parse wall-clock stamps of the first and
    the last update from a wrapper that timestamps the child's output stream."""
    cmd = [sys.executable, os.path.join(ROOT, "train_q_network.py"), folder, "-g", "0", "-d"]
    t_start = time.perf_counter()
    p = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, bufsize=0, env=dict(os.environ, PYTHONUNBUFFERED="1"))
    stamps = {}
    buf = b""
    tail = []
    while True:
        chunk = p.stdout.read(4096)
        if not chunk:
            break
        now = time.perf_counter()
        buf += chunk
        parts = buf.replace(b"\n", b"\r").split(b"\r")
        buf = parts[-1]
        for line in parts[:-1]:
            if line.startswith(b"batch:"):
                try:
                    n = int(line[6:line.index(b"/")])
                    stamps.setdefault(n, now)
                except ValueError:
                    pass
            elif line.strip():
                tail.append(line.decode(errors="replace"))
    rc = p.wait()
    return rc, t_start, time.perf_counter(), stamps, tail[-15:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trainer_e2e.json"))
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--loader-steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--frames", type=int, default=16384)
    ap.add_argument("--samples", type=int, default=65536)
    ap.add_argument("--work", default="/tmp/vdqn_e2e")
    args = ap.parse_args()
    from tools.bench_loader import make_shards
    shards = os.path.join(args.work, "shards")
    if not os.path.exists(os.path.join(shards, "index.npz")):
        make_shards(shards, n_frames=args.frames, shard_frames=1024, n_samples=args.samples)
    rec = {"batch": args.batch, "dataset": {"frames": args.frames, "samples": args.samples, "kind": "synthetic uint8 shards"}, "runs": {}}
    for tag, resident, steps, workers, shard_input in (("resident", True, args.steps, 0, "stream"), ("loader", False, args.steps, 0, "stream"),
                                                       ("dataloader", False, args.loader_steps, args.workers, "dataloader")):
        folder = os.path.join(args.work, f"exp_{tag}")
        make_cfg(folder, shards, steps, resident, workers, args.batch, shard_input)
        rc, t0, t1, stamps, tail = timed_run(folder)
        if rc != 0 or len(stamps) < 10:
            rec["runs"][tag] = {"rc": rc, "tail": tail}
            continue
        ns = sorted(stamps)
        # the printed counter of update n appears when update n has been QUEUED and update n-1's loss has been read back: the
        # interval between the stamps of two updates far apart is device time (the loss ring keeps the host one update ahead)
        warm = max(ns[0], min(ns[-1] - 10, steps // 5))
        first = min(n for n in ns if n >= warm)
        last = ns[-1]
        dt = stamps[last] - stamps[first]
        rec["runs"][tag] = {"rc": 0, "steps": steps, "workers": workers, "timed_updates": last - first, "seconds": round(dt, 4),
                            "ms_per_update": round(1e3 * dt / (last - first), 4), "tuples_per_s": round(args.batch * (last - first) / dt, 1),
                            "startup_seconds_to_first_update": round(stamps[ns[0]] - t0, 2), "process_seconds": round(t1 - t0, 2)}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--no-live-pmc"], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if line:
        b = json.loads(line[-1])
        rec["bench"] = {"tuples_per_s": b["value"], "ms_per_step": b["ms_per_step"]}
        for tag in ("resident", "loader"):
            if tag in rec["runs"] and rec["runs"][tag].get("tuples_per_s"):
                rec[f"{tag}_over_bench"] = round(rec["runs"][tag]["tuples_per_s"] / b["value"], 4)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
