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
    env.update(kw)
    return env


def _bench(args, **env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(**env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--ramp-seconds", "0", "--batch", "8", "--steps", "3", "--warmup", "1", "--profile-steps", "1", "--no-cpu-baseline", "--pool", "2"]


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
    """One rank, backend nccl (= RCCL): the three async all-reduces per update and their stream hand-off (the engine's
    side stream joined into torch's stream before each bucket is launched) run on the GPU; the result must equal the
    run without any collective."""
    a = _bench(["--gpus", "1", "--force-dist"] + SMALL)
    b = _bench(["--gpus", "1"] + SMALL)
    assert a["allreduce"]["backend"].startswith("nccl") and len(a["allreduce"]["buckets_bytes"]) == 3
    assert b["allreduce"] is None
    assert a["loss"] == pytest.approx(b["loss"], rel=2e-2)  # same data, same updates (bf16 + atomics: not bit-equal)


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
