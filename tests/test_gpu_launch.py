"""bench.py / train_q_network.py as the driver calls them — no torchrun on the command line — on the real HIP path:
the N > 1 path self-launches (two ranks share this box's one GPU, gloo carries the exchange: RCCL refuses two ranks on
one device), RCCL itself is exercised with one rank (--force-dist), and with two GPUs present the production exchange
(backend nccl) is checked against the single-process big-batch run."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.setdefault("VDQN_BENCH_NO_LIVE_PMC", "1")  # (the rocprofv3 child passes behind roofline.traffic have a test of their own)
    env.update(kw)
    return env


def _bench(args, **env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(**env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--ramp-seconds", "0", "--batch", "8", "--steps", "3", "--warmup", "1", "--windows", "1", "--profile-steps", "1", "--no-cpu-baseline", "--pool", "2"]


def test_per_bucket_adam_under_an_exchange_is_the_same_update():
    """VDQN_DIST_EARLY_ADAM=1 (round-4 review 7a): with a gradient exchange, stage 0 / stage 1 are updated behind THEIR bucket on a stream
    of their own instead of behind the last bucket.  One rank through RCCL, deterministic f32: the master parameters after three
    updates carry the same SHA-256 as without the switch and as the run without any exchange."""
    det = ["--dtype", "f32", "--deterministic", "--params-digest", "--batch", "4", "--steps", "3", "--warmup", "0", "--windows", "1", "--ramp-seconds", "0",
           "--profile-steps", "1", "--no-cpu-baseline", "--pool", "1"]
    a = _bench(["--gpus", "1", "--force-dist"] + det, VDQN_DIST_EARLY_ADAM="1")
    b = _bench(["--gpus", "1", "--force-dist"] + det)
    c = _bench(["--gpus", "1"] + det)
    assert a["params_sha256"] is not None and a["params_sha256"] == b["params_sha256"] == c["params_sha256"]
    assert a["loss"] == b["loss"] == c["loss"]


def test_bench_self_launches_two_ranks_without_torchrun():
    out = _bench(["--gpus", "2", "--backend", "gloo"] + SMALL, VDQN_BENCH_SINGLE_DEVICE="1")
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    assert len(out["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in out["per_rank_ms_per_step"])
    assert out["ms_per_step"] == pytest.approx(max(out["per_rank_ms_per_step"]), rel=1e-3)
    b = out["allreduce"]["buckets_bytes"]
    assert len(b) == 3 and sum(b) == 4 * 12426384  # the whole flat f32 gradient, in three stage buckets
    assert out["value"] > 0 and out["roofline"] is not None and out["cpu_baseline"] is None
    assert out["loss"] == out["loss"]  # finite


def test_bench_single_rank_through_rccl():
    """One rank, backend nccl (= RCCL): the three async all-reduces per update run on the GPU, each issued with the engine's
    gradient stream current (vdqn_net_grad_stream: where the stage's weight gradients and unfold kernel ran) and joined into the
    caller's stream by BucketAllReduce.finish() before Adam.  A one-rank SUM all-reduce is the identity, so in DETERMINISTIC f32
    mode the run with the collectives must leave the SAME BITS in the master parameters as the run without them: any
    mis-ordering of side stream -> RCCL -> Adam (a bucket reduced before its unfold finished, Adam before a bucket landed)
    changes them."""
    det = ["--dtype", "f32", "--deterministic", "--params-digest", "--batch", "4", "--steps", "3", "--warmup", "0", "--windows", "1", "--ramp-seconds", "0",
           "--profile-steps", "1", "--no-cpu-baseline", "--pool", "1"]
    a = _bench(["--gpus", "1", "--force-dist"] + det)
    b = _bench(["--gpus", "1"] + det)
    assert a["allreduce"]["backend"].startswith("nccl") and len(a["allreduce"]["buckets_bytes"]) == 3
    assert sum(a["allreduce"]["buckets_bytes"]) == 4 * 12426384
    assert b["allreduce"] is None and a["deterministic"] and b["deterministic"]
    assert a["params_sha256"] is not None and a["params_sha256"] == b["params_sha256"]
    assert a["loss"] == b["loss"]
    # and the throughput mode (bf16, atomic sums): same data, same updates, not bit-equal
    a = _bench(["--gpus", "1", "--force-dist"] + SMALL)
    b = _bench(["--gpus", "1"] + SMALL)
    # (atomic sums + bf16 at batch 4: Adam turns rounding-level gradient elements into +-lr steps, two runs of the SAME command differ by
    # up to ~3 % in the loss after three updates.  The per-update statement is tests/test_gpu_engine.py::
    # test_default_schedule_gradients_equal_deterministic_per_tensor: one update of the default schedule equals deterministic mode to
    # 1e-5 (f32) / 1e-4 (bf16) relative L2 in every gradient tensor.)
    assert a["loss"] == pytest.approx(b["loss"], rel=0.1)


def test_bench_eight_ranks_on_one_device_gloo():
    """BASELINE config 4's rank count, functionally: eight self-launched ranks at batch 2 share this box's GPU (gloo carries the
    exchange); catches anything that depends on the number of ranks (bucket slicing, rank-strided seeds, the MAX over ranks)."""
    out = _bench(["--gpus", "8", "--backend", "gloo", "--ramp-seconds", "0", "--batch", "2", "--steps", "2", "--warmup", "1", "--no-profile",
                  "--no-cpu-baseline", "--pool", "1"], VDQN_BENCH_SINGLE_DEVICE="1")
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp8"
    assert len(out["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in out["per_rank_ms_per_step"])
    assert out["ms_per_step"] == pytest.approx(max(out["per_rank_ms_per_step"]), rel=1e-3)
    b = out["allreduce"]["buckets_bytes"]
    assert len(b) == 3 and sum(b) == 4 * 12426384
    assert out["loss"] == out["loss"] and out["value"] > 0


def test_c_abi_comm_one_rank(tmp_path):
    """include/vdqn.h's own exchange entries (vdqn_comm_unique_id / _init / vdqn_allreduce_bucket / _destroy) on RCCL with one
    rank: the all-reduce is the identity for f32 and bf16 buffers, bad arguments are errors, and a deterministic f32 update
    whose three stage buckets go through them (CAbiBucketAllReduce, each queued on vdqn_net_grad_stream) leaves the same bits as
    an update without any exchange."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from video_dqn_amd import _lib, synth
    from video_dqn_amd.dist import CAbiBucketAllReduce
    from video_dqn_amd.engine import NetEngine, TDStepper
    lib = _lib.load()
    comm = CAbiBucketAllReduce(0, 1, str(tmp_path / "uid"), force=True)
    assert lib.vdqn_comm_rank(comm.handle) == 0 and lib.vdqn_comm_size(comm.handle) == 1
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(1 << 20, device="cuda")
    y = x.clone()
    _lib.check(lib.vdqn_allreduce_bucket(comm.handle, x.data_ptr(), x.numel(), _lib.VDQN_F32, st), "allreduce f32")
    xb = torch.randn(4096, device="cuda").to(torch.bfloat16)
    yb = xb.clone()
    _lib.check(lib.vdqn_allreduce_bucket(comm.handle, xb.data_ptr(), xb.numel(), _lib.VDQN_BF16, st), "allreduce bf16")
    torch.cuda.synchronize()
    assert torch.equal(x, y) and torch.equal(xb, yb)
    assert lib.vdqn_allreduce_bucket(comm.handle, x.data_ptr(), 16, 5, st) < 0 and b"dtype" in lib.vdqn_last_error()
    assert lib.vdqn_allreduce_bucket(None, x.data_ptr(), 16, 0, st) < 0
    h = C.c_void_p()
    assert lib.vdqn_comm_init(2, 2, C.create_string_buffer(128), C.byref(h)) < 0  # rank out of range: refused before RCCL is asked

    def run(with_comm):
        B = 4
        net = NetEngine(3, 5, 1, True, "f32", 2 * B, deterministic=True)
        net.load_tensors(synth.make_state_dict(7))
        stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, allreduce=(comm.launch if with_comm else None))
        for step in (1, 2, 3):
            (tup, _) = synth.make_batch(300 + step, B, 1, structured=True, reward_p=0.3)
            stp.step(tup[0].contiguous().cuda(), tup[1].contiguous().cuda(), 1, tup[2].cuda(), tup[3].float().cuda(), tup[4].float().cuda(),
                     finish_allreduce=(comm.finish if with_comm else None))
        torch.cuda.synchronize()
        return net.params.cpu().clone()
    pa, pb = run(True), run(False)
    assert comm.bucket_bytes and sum(comm.bucket_bytes) == 4 * 12426384
    assert torch.equal(pa, pb)
    comm.close()


def test_bench_failed_rank_is_a_failed_job():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "no_such_backend"] + SMALL,
                       env=_env(VDQN_BENCH_SINGLE_DEVICE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_train_cli_self_launches_two_ranks(tmp_path):
    folder = tmp_path / "exp"
    folder.mkdir()
    (folder / "config.yml").write_text(
        "DATASET: 'synthetic'\nPANORAMA: False\nLOSS_CLIP: 'rect'\nARCHITECTURE: 'extra_capacity'\nLEARNING_RATE: 0.0001\n"
        "GAMMA: 0.99\nCHECKPOINT_INTERVAL: 2\nNUM_STEPS: 2\nTARGET_UPDATE_INTERVAL: 2\nSEED: 4\nBATCH_SIZE: 4\nNUM_WORKERS: 0\n"
        "COMPUTE_DTYPE: 'f32'\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_q_network.py"), str(folder), "-g", "0,0"],
                       env=_env(VDQN_DIST_BACKEND="gloo", VDQN_SINGLE_DEVICE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    snap = torch.load(folder / "models" / "sample2.torch", map_location="cpu")
    assert snap["sample_number"] == 2 and snap["optimizer_state_dict"]["state"][0]["step"] == 2
    assert "RANDOM initialisation" in r.stdout + r.stderr  # no PRETRAINED_WEIGHTS: the trainer says so


# ---- two real GPUs: the production exchange over RCCL ------------------------------------------------------------
def _worker_nccl(rank, world, port, out_dir, arch_ec):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from video_dqn_amd import synth
    from video_dqn_amd.dist import BucketAllReduce
    from video_dqn_amd.engine import NetEngine, TDStepper
    B = 4
    net = NetEngine(3, 5, 1, arch_ec, "f32", 2 * B, device=dev)
    net.load_tensors(synth.make_state_dict(7, extra_capacity=arch_ec))
    comm = BucketAllReduce(world)
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, world_size=world, allreduce=comm.launch)
    if not arch_ec:
        net.set_bn_sync(world)  # SyncBN through dist.all_reduce called from inside the C call
    for step in (1, 2):
        (tup, _) = synth.make_batch(200 + step, 2 * B, 1, structured=True, reward_p=0.3)
        lo, hi = rank * B, rank * B + B
        stp.step(tup[0][lo:hi].contiguous().to(dev), tup[1][lo:hi].contiguous().to(dev), 1, tup[2][lo:hi].to(dev),
                 tup[3][lo:hi].float().to(dev), tup[4][lo:hi].float().to(dev), finish_allreduce=comm.finish)
    torch.cuda.synchronize()
    torch.save({"params": net.params.cpu(), "bnstats": net.bnstats.cpu()}, os.path.join(out_dir, f"nccl{int(arch_ec)}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _worker_cabi(rank, world, uid_path, out_dir):
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    from video_dqn_amd import synth
    from video_dqn_amd.dist import CAbiBucketAllReduce
    from video_dqn_amd.engine import NetEngine, TDStepper
    B = 4
    comm = CAbiBucketAllReduce(rank, world, uid_path)
    net = NetEngine(3, 5, 1, True, "f32", 2 * B, device=dev)
    net.load_tensors(synth.make_state_dict(7))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, world_size=world, allreduce=comm.launch)
    for step in (1, 2):
        (tup, _) = synth.make_batch(200 + step, 2 * B, 1, structured=True, reward_p=0.3)
        lo, hi = rank * B, rank * B + B
        stp.step(tup[0][lo:hi].contiguous().to(dev), tup[1][lo:hi].contiguous().to(dev), 1, tup[2][lo:hi].to(dev),
                 tup[3][lo:hi].float().to(dev), tup[4][lo:hi].float().to(dev), finish_allreduce=comm.finish)
    torch.cuda.synchronize()
    torch.save({"params": net.params.cpu()}, os.path.join(out_dir, f"cabi_{rank}.pt"))
    comm.close()


def test_two_gpus_c_abi_comm_equal_one_big_batch(tmp_path):
    """The exchange through include/vdqn.h's own RCCL entries (no torch.distributed) on two GPUs == one big batch."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    from video_dqn_amd import synth
    from video_dqn_amd.engine import NetEngine, TDStepper
    mp.spawn(_worker_cabi, args=(2, str(tmp_path / "uid"), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "cabi_0.pt"), torch.load(tmp_path / "cabi_1.pt")
    assert torch.equal(r0["params"], r1["params"])
    B = 8
    net = NetEngine(3, 5, 1, True, "f32", 2 * B)
    net.load_tensors(synth.make_state_dict(7))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    for step in (1, 2):
        (tup, _) = synth.make_batch(200 + step, B, 1, structured=True, reward_p=0.3)
        stp.step(tup[0].contiguous().cuda(), tup[1].contiguous().cuda(), 1, tup[2].cuda(), tup[3].float().cuda(), tup[4].float().cuda())
    torch.cuda.synchronize()
    nt = net.trainable_numel
    d = (net.params.cpu()[:nt] - r0["params"][:nt]).abs()
    assert d.max().item() <= 2.5e-4 and d.mean().item() < 2e-6


@pytest.mark.parametrize("arch_ec", [True, False], ids=["extra_capacity", "basic_syncbn"])
def test_two_gpus_rccl_equal_one_big_batch(tmp_path, arch_ec):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    from test_gpu_ddp import _free_port
    from video_dqn_amd import synth
    from video_dqn_amd.engine import NetEngine, TDStepper
    mp.spawn(_worker_nccl, args=(2, _free_port(), str(tmp_path), arch_ec), nprocs=2, join=True)
    r0 = torch.load(tmp_path / f"nccl{int(arch_ec)}_0.pt")
    r1 = torch.load(tmp_path / f"nccl{int(arch_ec)}_1.pt")
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["bnstats"], r1["bnstats"])
    B = 8
    net = NetEngine(3, 5, 1, arch_ec, "f32", 2 * B)
    net.load_tensors(synth.make_state_dict(7, extra_capacity=arch_ec))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    for step in (1, 2):
        (tup, _) = synth.make_batch(200 + step, B, 1, structured=True, reward_p=0.3)
        stp.step(tup[0].contiguous().cuda(), tup[1].contiguous().cuda(), 1, tup[2].cuda(), tup[3].float().cuda(), tup[4].float().cuda())
    torch.cuda.synchronize()
    nt = net.trainable_numel
    d = (net.params.cpu()[:nt] - r0["params"][:nt]).abs()
    assert d.max().item() <= 2.5e-4 and d.mean().item() < (2e-6 if arch_ec else 2e-5)


def test_bench_roofline_traffic_is_measured_live():
    """bench.py's `roofline.traffic` comes from two rocprofv3 --pmc child passes of the same command started before the process
    touches the GPU (FETCH_SIZE and WRITE_SIZE in separate passes, MI355X_MICROARCH.md): present, labelled live, and of the order of
    the kernel's algorithmic bytes (0.5x .. 4x).  Skipped where rocprofv3 is absent."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("no rocprofv3 on this box")
    out = _bench(["--ramp-seconds", "0", "--batch", "32", "--steps", "3", "--warmup", "1", "--profile-steps", "1", "--no-cpu-baseline", "--pool", "1"],
                 VDQN_BENCH_NO_LIVE_PMC="0")
    r = out["roofline"]
    assert r is not None and r["traffic"] is not None and r["traffic_source"].startswith("live"), r
    assert 0.5 * r["alg_bytes_per_launch"] <= r["traffic"] <= 4.0 * r["alg_bytes_per_launch"], r
