#!/usr/bin/env python
"""Benchmark of the hot path: (s,a,r,s') TD-updates/s of one full Q-learning update
(online forward over [s; s'], target forward over s', Double-DQN target + TD loss, backward, Adam) on
synthetic 224x224 RGB frames — BASELINE.json's metric on its config[1] (ResNet-18 HabitatDQNMultiAction,
5 categories x 3 actions, batch 256 per GPU, bf16, extra_capacity / real_data hyper-parameters).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU; for N > 1 the flat gradient is all-reduced over RCCL in three buckets as the staged
backward completes them (weak scaling: 256 samples per GPU).  Rank 0 prints ONE JSON line.

Called WITHOUT a launcher, ``--gpus N`` (N > 1) starts the N rank processes itself: the parent does so before any
GPU call (it never initialises HIP), relays rank 0's JSON line and exits with the ranks' code
(video_dqn_amd/launch.py).  ``VDQN_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo`` runs the N > 1
path on a 1-GPU box (functional check only: the ranks share the device).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime initialises: see video_dqn_amd/__init__.py

import torch  # noqa: E402

GFLOP_PER_TUPLE_F1 = 17.982854656  # 3 fwd + 1 bwd, SURVEY.md §8(d) (8,991,427,328 MAC)
PEAK_BF16_TFLOPS = 2500.0           # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-seconds", type=float, default=1.5,
                    help="untimed updates run before the warm-up steps so that the device holds its steady clocks (0 = none)")
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU")
    ap.add_argument("--frames", type=int, default=1, help="views per sample (12 = BASELINE config 5)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--arch", default="extra_capacity", choices=["extra_capacity", "basic"],
                    help="ARCHITECTURE key (north_star / real_data config: extra_capacity; defaults.py: basic)")
    ap.add_argument("--target-update-interval", type=int, default=1000)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--h2d", default="none", choices=["none", "serial", "overlap"],
                    help="PCIe-inclusive variant (never the headline value): copy the uint8 frames from pinned host memory every step, "
                         "on the compute stream (serial) or double-buffered on a copy stream (overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-profile", action="store_true", help="skip the event-profiled steps (roofline = null)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the two rocprofv3 --pmc child passes for roofline.traffic "
                                                                 "(the committed passes of profiles/pmc_latest.json are quoted instead)")
    ap.add_argument("--pack-ahead", action="store_true",
                    help="pack the next minibatch's frames under the current update's backward pass (default: every update packs its own frames at its start)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5,
                    help="consecutive timed windows of --steps updates each (barrier + synchronize around every one); `value` is the MEDIAN "
                         "window, `window_values` lists them all (1 = the single window of rounds 1-4)")
    ap.add_argument("--pool", type=int, default=4,
                    help="distinct synthetic minibatches resident in HBM, used round-robin (4 x 77 MB of frames at batch 256 do "
                         "not fit the 256 MiB Infinity Cache together, so no step reads its frames from cache)")
    ap.add_argument("--loss-kind", default="l2", choices=["l2", "huber"], help="l2 = the reference's loss (train_q_network.py:167)")
    ap.add_argument("--c4", action="store_true",
                    help="BASELINE config 4's timing window: at least 2200 timed steps so that two target refreshes "
                         "(every 1000 updates) fall inside it")
    ap.add_argument("--deterministic", action="store_true",
                    help="DETERMINISTIC mode (ordered split-K sums, one-block loss): run-to-run bit-identical updates")
    ap.add_argument("--params-digest", action="store_true",
                    help="add `params_sha256` (SHA-256 of the master parameters after the last timed update, rank 0) to the line; "
                         "with --deterministic and --pool 1 two runs from the same seed must agree bit for bit")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and issue the bucketed all-reduces even with one rank (RCCL and the "
                         "stream ordering of the exchange exercised on a single GPU)")
    args = ap.parse_args()
    if args.c4:
        args.steps = max(args.steps, 2200)
        args.target_update_interval = 1000
        args.windows = 1
    args.windows = max(1, args.windows)
    return args


def cpu_baseline(batch: int, budget_s: float = 25.0):
    """The oracle (oracle/ref_cpu.py, a torch-CPU fp32 restatement of the reference path) timed on this box's
    host cores: once with torch.set_num_threads(1) as the reference itself runs (train_q_network.py:85) and
    once with the cores this process may use (capped at 32 threads: more only adds oversubscription at this
    batch size).  Reported, never shipped: this is the only place bench.py touches oracle/."""
    from oracle import ref_cpu
    from video_dqn_amd import synth
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    tr = ref_cpu.Trainer(ref_cpu.default_config(), synth.make_state_dict(7))
    (tup, _) = synth.make_batch(1, batch, 1)

    def timed(threads, budget):
        torch.set_num_threads(threads)
        tr.step(tup)  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            tr.step(tup)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget or n >= 8:
                return n, dt / n

    n1, dt1 = timed(1, budget_s * 0.4)
    threads = max(1, min(avail, 32))
    nn, dtn = timed(threads, budget_s * 0.4)
    best_threads, best_dt = (threads, dtn) if dtn < dt1 else (1, dt1)
    return {"value": round(batch / best_dt, 3), "unit": "tuples/s", "cores": best_threads, "kind": "port",
            "sample": f"{n1}+{nn} full TD updates at batch {batch} (fp32 torch-CPU oracle of the reference path; "
                      f"1 thread = the reference's own setting, and {threads} threads)",
            "single_thread_value": round(batch / dt1, 3), "multi_thread_value": round(batch / dtn, 3),
            "multi_threads": threads, "host_cores_visible": avail}


def live_pmc_traffic(args):
    """HBM bytes per launch of every kernel, measured in THIS run: two rocprofv3 child passes (--kernel-trace --pmc FETCH_SIZE, then
    --pmc WRITE_SIZE: separate passes, no other tracing, the program itself behind `--`, as MI355X_MICROARCH.md prescribes) of this
    bench command at 2 steps with the side streams serialised.  They run AFTER the timed region and the event-profiled steps (this
    process then only holds its memory), each bounded by a 100 s timeout; the scratch directory is removed whatever happens.
    Corrections as in tools/pmc_summary.py (FETCH_SIZE counts half of wide streaming reads on gfx950).
    Returns ({kernel tag: bytes per launch} or None, reason): None = the caller quotes the committed passes and says why."""
    import shutil
    import subprocess
    import tempfile
    if args.no_live_pmc or os.environ.get("VDQN_BENCH_NO_LIVE_PMC") == "1":
        return None, "switched off (--no-live-pmc / VDQN_BENCH_NO_LIVE_PMC=1)"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return None, "this process is itself being profiled"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    tmp = None
    try:
        from pmc_summary import per_kernel
        tmp = tempfile.mkdtemp(prefix="vdqn_pmc_", dir="/tmp")
        tot = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", os.path.join(tmp, counter), "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--windows", "1", "--ramp-seconds", "0", "--no-cpu-baseline",
                   "--no-profile", "--no-live-pmc", "--pool", "1", "--batch", str(args.batch), "--frames", str(args.frames), "--dtype", args.dtype,
                   "--arch", args.arch, "--loss-kind", args.loss_kind]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", VDQN_NO_OVERLAP="1"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, timeout=100)
            except subprocess.TimeoutExpired:
                return None, f"the {counter} pass exceeded its 100 s timeout"
            if r.returncode != 0:
                tail = (r.stderr or b"").decode(errors="replace").strip().splitlines()[-1:] or [""]
                return None, f"the {counter} pass returned {r.returncode}: {tail[0][:160]}"
            csvs = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(tmp, counter)) for f in fs if f.endswith("counter_collection.csv")]
            if not csvs:
                return None, f"the {counter} pass wrote no counter_collection.csv"
            tot[counter] = per_kernel(csvs[0], counter)
        (ft, fc), (wt, wc) = tot["FETCH_SIZE"], tot["WRITE_SIZE"]
        return {t: int(round(2 * 1024 * ft[t] / max(fc[t], 1) + 1024 * wt[t] / max(wc[t], 1))) for t in set(ft) | set(wt)}, "live"
    except Exception as e:  # noqa: BLE001 - any failure falls back to the committed passes, with the reason on the line
        return None, f"{type(e).__name__}: {e}"[:200]
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(kernel: str, batch: int, frames: int, dtype: str, arch: str = "extra_capacity"):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_latest.json: rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as the gfx950 guide prescribes).
    PMC collection cannot run inside this process; the figure is only quoted for the configuration it was
    collected on (batch 256, 1 frame, bf16), otherwise null."""
    if (batch, frames, dtype, arch) != (256, 1, "bf16", "extra_capacity"):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            return json.load(f)["per_launch"][kernel]["traffic_bytes"]
    except Exception:
        return None


def main():
    args = parse()
    from video_dqn_amd import launch
    launch.die_with_parent()  # a rank started by spawn_ranks goes down with its launcher
    if args.gpus > 1 and not launch.in_rank_env():
        # no launcher around us: become one.  Nothing above has touched the GPU (importing torch does not), so the
        # children are started from a HIP-free parent that only waits for them.
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    out_stream = launch.claim_stdout()  # the JSON line goes here; fd 1 now points at stderr (native-library chatter)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if os.environ.get("VDQN_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0  # functional test of the N > 1 path on a 1-GPU box (use --backend gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(launch.free_port())
        dist.init_process_group(args.backend, rank=rank, world_size=world, **({"device_id": dev} if args.backend == "nccl" else {}))

    from video_dqn_amd import _lib, synth
    from video_dqn_amd.engine import NetEngine, TDStepper
    from video_dqn_amd.dist import BucketAllReduce

    B, F = args.batch, args.frames
    ec = args.arch == "extra_capacity"
    net = NetEngine(3, 5, F, ec, args.dtype, 2 * B, device=dev, deterministic=(True if args.deterministic else None))
    net.load_tensors(synth.make_state_dict(4, extra_capacity=ec, num_frames=F))  # same seed on every rank: replicas start identical
    comm = BucketAllReduce(world, force=args.force_dist) if use_dist else None
    if world > 1 and not ec:
        net.set_bn_sync(world)  # 'basic': global BatchNorm statistics (N ranks == one big batch)
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, target_update_interval=args.target_update_interval,
                    world_size=world, allreduce=(comm.launch if comm else None), loss_kind=args.loss_kind,
                    allreduce_wait=(comm.wait_last if comm else None))

    # synthetic minibatches, resident in HBM before the timed region: uint8 frames (normalise fused into packing).
    # `--pool` distinct ones are used round-robin so that a step does not find its frames in the Infinity Cache.
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    n_pool = max(1, args.pool if args.h2d == "none" else 1)
    pool = []
    for _ in range(n_pool):
        b_ = torch.randint(0, 256, (B, F, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
        a_ = torch.randint(0, 256, (B, F, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
        act_ = torch.randint(0, 3, (B,), dtype=torch.int64, device=dev, generator=g)
        rew_ = (torch.rand((B, 5), device=dev, generator=g) < 0.05).float()
        pool.append((b_, a_, act_, rew_, rew_.clone()))
    before, after, act, rew, term = pool[0]
    pool_i = {"i": 0}

    if args.h2d != "none":
        host = [t.cpu().pin_memory() for t in (before, after)]
        bufs = [(before, after), (torch.empty_like(before), torch.empty_like(after))]
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [None, None]
        state = {"i": 0}

        def prefetch(slot):
            with torch.cuda.stream(copy_stream):
                bufs[slot][0].copy_(host[0], non_blocking=True)
                bufs[slot][1].copy_(host[1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            ready[slot] = ev

    def one_step():
        if args.h2d == "serial":
            before.copy_(host[0], non_blocking=True)
            after.copy_(host[1], non_blocking=True)
        elif args.h2d == "overlap":
            i = state["i"]
            if ready[i & 1] is None:
                prefetch(i & 1)
            torch.cuda.current_stream().wait_event(ready[i & 1])
            # the other buffer was consumed by the previous step's kernels already queued on the compute stream
            copy_stream.wait_stream(torch.cuda.current_stream())
            prefetch((i + 1) & 1)
            state["i"] = i + 1
            b, a = bufs[i & 1]
            return stp.step(b, a, 0, act, rew, term, finish_allreduce=(comm.finish if comm else None))
        b_, a_, act_, rew_, term_ = pool[pool_i["i"] % n_pool]
        pool_i["i"] += 1
        # --pack-ahead: the next minibatch's frames are packed under this update's backward pass instead of at the start of the next
        # update (TDStepper.step, next_frames).  Measured 5.806 against 5.739 ms per update (profiles/r03w_ab_pack_ahead.txt): the
        # HBM-bound pack beside layer4 and the head costs more than the start of the update gains.  Off by default.
        nxt = None
        if args.pack_ahead or os.environ.get("VDQN_BENCH_PACK_AHEAD") == "1":
            nb_, na_ = pool[pool_i["i"] % n_pool][:2]
            nxt = (nb_, na_, 0, True)  # (pool tensors are never rewritten: an announcement that aliases this step's frames is a replay)
        return stp.step(b_, a_, 0, act_, rew_, term_, finish_allreduce=(comm.finish if comm else None), next_frames=nxt)

    # device ramp (untimed, before the W warm-up steps): a box that sat idle takes longer than 20 updates (0.13 s) to reach the
    # clocks it then holds; the same updates are repeated for --ramp-seconds so that short runs (--steps 20) are not timed on
    # the ramp (profiles/r02p_warmup_ab.txt).  Every rank runs the same number of updates (they contain the collectives).
    ramp_steps = 0
    if args.ramp_seconds > 0:
        t_r = time.perf_counter()
        for _ in range(10):
            one_step()
        torch.cuda.synchronize()
        per = max((time.perf_counter() - t_r) / 10, 1e-4)
        ramp_steps = 10 + max(0, int(args.ramp_seconds / per) - 10)
        if world > 1:
            t = torch.tensor([ramp_steps], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ramp_steps = int(t.item())
        for _ in range(ramp_steps - 10):
            one_step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sample0 = stp.sample_number
    # --windows consecutive timed windows of EXACTLY --steps updates, each bracketed by barrier + synchronize on both sides and
    # reduced by MAX over ranks; the reported value is the median window (a single 20-step window is 0.11 s: inside the box's noise)
    win_elapsed, win_rank_ms = [], []
    for _w in range(args.windows):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rank_ms = [round(1e3 * el / args.steps, 3)]
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            rank_ms = [round(1e3 * x.item() / args.steps, 3) for x in allt]
            el = max(x.item() for x in allt)  # MAX over ranks
        win_elapsed.append(el)
        win_rank_ms.append(rank_ms)
    order = sorted(range(args.windows), key=lambda i: win_elapsed[i])
    med = order[(args.windows - 1) // 2]  # (an even count: the slower of the two middle windows)
    elapsed, per_rank_ms = win_elapsed[med], win_rank_ms[med]
    loss_val = float(loss.item())
    params_sha = None
    if args.params_digest and rank == 0:
        import hashlib
        params_sha = hashlib.sha256(net.params.detach().cpu().numpy().tobytes()).hexdigest()
    refreshes = (stp.sample_number // args.target_update_interval) - (sample0 // args.target_update_interval)

    # ---- live per-kernel timing (HIP events on the launch stream) for the roofline of the dominant kernel ----
    roofline = None
    kernels = None
    if not args.no_profile:
        # per-kernel roofline timing: serialise the engine's side stream so every kernel has the chip to itself
        # (in the timed region above weight gradients / the target forward overlap the main chain).  Every rank
        # runs these steps (they contain the gradient all-reduce); only rank 0 records and reports.
        net.lib.vdqn_net_set_overlap(net.handle, 0)
        torch.cuda.synchronize()
        per_step = []  # one profile per event-profiled step (rank 0): the spread of the dominant kernel's launch time over them
        for _ in range(args.profile_steps):
            if rank == 0:
                _lib.profile_enable(True)
            stp.forward_backward(before, after, 0, act, rew, term)
            if comm:
                comm.finish()
            stp.optimizer_step()
            torch.cuda.synchronize()
            if rank == 0:
                per_step.append(_lib.profile_collect())
                _lib.profile_enable(False)
        net.lib.vdqn_net_set_overlap(net.handle, 1)
    prof = None
    if not args.no_profile and rank == 0:
        prof = {}
        for ps in per_step:
            for k, v in ps.items():
                d = prof.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                for f in ("launches", "ms", "flops", "bytes"):
                    d[f] += v[f]
        tot_ms = sum(v["ms"] for v in prof.values())
        kernels = {k: {"launches_per_step": v["launches"] // args.profile_steps, "ms_per_step": round(v["ms"] / args.profile_steps, 4),
                       "share": round(v["ms"] / tot_ms, 4),
                       "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["flops"] > 0 else None,
                       "alg_gbs": round(v["bytes"] / v["ms"] / 1e6, 1)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

    # the shader clock this device holds under a full-rate bf16 MFMA stream on random operands (in-kernel s_memtime /
    # s_memrealtime): what the 2.5 PFLOP/s nominal peak of `roofline.peak` shrinks to on this box
    # (measured right behind the profiled steps, while the device is still hot)
    mfma_clock = None
    if rank == 0 and not args.no_profile:
        try:
            import ctypes as C
            raw = C.CDLL(_lib.LIB_PATH)
            ghz, tf = C.c_double(0.0), C.c_double(0.0)
            if raw.vdqn_debug_mfma_clock(C.byref(ghz), C.byref(tf)) == 0:
                mfma_clock = {"shader_clock_ghz": round(ghz.value, 3), "bare_mfma_tflops": round(tf.value, 1),
                              "how": "~11 ms of back-to-back v_mfma_f32_32x32x16_bf16 on random bf16 operands after 45 ms of heat, one wave per SIMD; "
                                     "clock = delta s_memtime / delta s_memrealtime x 100 MHz, median over waves"}
        except Exception:
            mfma_clock = None

    if prof:
        name, v = max(prof.items(), key=lambda kv: kv[1]["ms"])
        step_us = sorted(1e3 * ps[name]["ms"] / ps[name]["launches"] for ps in per_step if name in ps and ps[name]["launches"])
        spread = {"min": round(step_us[0], 2), "median": round(step_us[(len(step_us) - 1) // 2], 2), "max": round(step_us[-1], 2),
                  "over": f"{len(step_us)} profiled steps (average launch of each)"} if step_us else None
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        if v["flops"] > 0:
            # roofline.traffic, live: two rocprofv3 --pmc child passes of this command, AFTER everything that is timed (one process,
            # the default path only); the committed passes are quoted, with the reason, when they are skipped or fail
            live_traffic, live_reason = None, "not a default one-process run"
            if world == 1 and not args.force_dist and args.h2d == "none":
                live_traffic, live_reason = live_pmc_traffic(args)
            ach = v["flops"] / v["ms"] / 1e9
            live = (live_traffic or {}).get(name)
            if live_traffic is not None and not live:
                live_reason = f"the live passes hold no row for {name}"
            roofline = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4),
                        # the same achieved rate against what THIS box's matrix pipes delivered a moment later on random operands with
                        # nothing else to do (mfma_clock_under_load.bare_mfma_tflops: the power-limited clock, not the 2.5 PFLOP/s nominal)
                        "frac_of_measured_bare": (round(ach / mfma_clock["bare_mfma_tflops"], 4) if (mfma_clock and args.dtype == "bf16"
                                                                                                    and mfma_clock["bare_mfma_tflops"] > 0) else None),
                        "traffic": live if live else pmc_traffic(name, B, F, args.dtype, args.arch),
                        "traffic_source": ("live: two rocprofv3 child passes of this command (--kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE; "
                                           "2 steps, side streams serialised) run after the timed region; 2 x FETCH_SIZE + WRITE_SIZE per launch"
                                           if live else
                                           "profiles/pmc_latest.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           f"command, committed, per launch (live passes: {live_reason})"),
                        "avg_launch_us": round(1e3 * v["ms"] / v["launches"], 2), "avg_launch_us_per_step": spread, "launches": v["launches"],
                        "alg_flops_per_launch": round(v["flops"] / v["launches"]),
                        "alg_bytes_per_launch": round(v["bytes"] / v["launches"]),
                        "measured": f"HIP events around every launch, {args.profile_steps} steps right after the timed region, side-stream overlap off"}
        else:
            ach = v["bytes"] / v["ms"] / 1e6
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                        "avg_launch_us": round(1e3 * v["ms"] / v["launches"], 2), "avg_launch_us_per_step": spread, "launches": v["launches"]}
    if rank == 0:
        tuples = B * world * args.steps
        value = tuples / elapsed
        gflop_tuple = GFLOP_PER_TUPLE_F1 * F
        out = {
            "metric": "(s,a,r,s') TD-updates/sec, 224x224 frames, batch 256, 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "tuples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ramp_steps": ramp_steps,
            "windows": args.windows, "window_values": [round(B * world * args.steps / e, 2) for e in win_elapsed],
            "window_ms_per_step": [round(1e3 * e / args.steps, 3) for e in win_elapsed],
            "value_is": f"median of {args.windows} consecutive windows of {args.steps} updates" if args.windows > 1 else f"one window of {args.steps} updates",
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if args.h2d == "none" else f"synthetic, uint8 frames copied from pinned host memory every step ({args.h2d})",
            "config": {"workload": f"HabitatDQNMultiAction ResNet-18 {args.arch}, 5 categories x 3 actions, full TD update "
                                   f"(online fwd on [s;s'], target fwd on s', Double-DQN target + {'L2' if args.loss_kind == 'l2' else 'Huber'} TD loss, backward, Adam)",
                       "batch_per_gpu": B, "global_batch": B * world, "frames_per_sample": F, "frame": "224x224x3 uint8 (normalise fused)",
                       "parallelism": f"dp{world}", "target_update_interval": args.target_update_interval,
                       "gamma": 0.99, "loss_clip": "rect", "lr": 1e-4, "loss_kind": args.loss_kind,
                       "minibatch_pool": n_pool,
                       "frame_pack": ("the next minibatch's frames are packed under the current update's backward pass (inside the timed region)"
                                      if ((args.pack_ahead or os.environ.get("VDQN_BENCH_PACK_AHEAD") == "1") and args.h2d == "none")
                                      else "at the start of every update")},
            "per_rank_ms_per_step": per_rank_ms,
            "target_refreshes_in_window": refreshes,
            "allreduce": ({"backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""), "buckets_bytes": comm.bucket_bytes,
                           "overlapped_with": "backward stages 1-2 (bucket s is launched when stage s's unfold_grads is queued)"}
                          if comm else None),
            "model_tflops": round(value * gflop_tuple / 1e3, 2),
            "model_frac_of_bf16_peak": round(value * gflop_tuple / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
            "loss": loss_val,
            "deterministic": bool(net.deterministic),
            "params_sha256": params_sha,
            "roofline": roofline,
            "mfma_clock_under_load": mfma_clock,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), file=out_stream, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
