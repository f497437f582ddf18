"""Two ranks on ONE GPU (gloo carries the exchange; RCCL refuses two ranks per device) drive the real HIP path
through TDStepper + the bucketed all-reduce hook: after two updates both ranks hold the same parameters, equal to
a single-process run on the concatenated batch (DDP == one big batch)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(seed, B):
    from video_dqn_amd import synth
    (tup, raw) = synth.make_batch(seed, B, 1, structured=True, reward_p=0.3)
    return tup


def _run(net_stepper, tup, lo, hi, finish=None):
    net, stp = net_stepper
    dev = "cuda"
    before, after, act, rew, term = (tup[0][lo:hi], tup[1][lo:hi], tup[2][lo:hi], tup[3][lo:hi], tup[4][lo:hi])
    stp.step(before.contiguous().to(dev), after.contiguous().to(dev), 1, act.to(dev), rew.float().to(dev), term.float().to(dev),
             finish_allreduce=finish)
    torch.cuda.synchronize()


def _make(B, world, hook=None, deterministic=True):
    from video_dqn_amd import synth
    from video_dqn_amd.engine import NetEngine, TDStepper
    net = NetEngine(3, 5, 1, True, "f32", 2 * B, deterministic=deterministic)
    net.load_tensors(synth.make_state_dict(7))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, world_size=world, allreduce=hook)
    return net, stp


def _worker(rank, world, port, out_dir, per_rank=4, deterministic=True):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)

    def hook(grad_slice, stage):  # test transport: through the host, so it works whatever gloo's GPU support is
        torch.cuda.synchronize()
        h = grad_slice.cpu()
        dist.all_reduce(h)
        grad_slice.copy_(h)

    ns = _make(per_rank, world, hook, deterministic)
    for step in (1, 2):
        tup = _batch(200 + step, per_rank * world)
        _run(ns, tup, rank * per_rank, (rank + 1) * per_rank)
    torch.save({"params": ns[0].params.cpu(), "loss": ns[1].loss.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,per_rank,deterministic", [(2, 4, True), (4, 2, True), (8, 2, True), (2, 4, False)],
                         ids=["world2", "world4", "world8", "world2_default_atomic_mode"])
def test_n_ranks_equal_one_big_batch(tmp_path, world, per_rank, deterministic):
    """N ranks x per_rank samples == ONE process on the N * per_rank samples (the reference is one process,
    train_q_network.py:255-259,275): after two updates every replica holds bit-identical parameters (same reduced gradient,
    same Adam), and they equal the big-batch run's up to the summation order of the gradient (f32, deterministic mode).
    Arithmetic, not plumbing: world 4 and 8 run the same bound as world 2.  One case runs the SHIPPED default (f32 atomic sums
    for the split-K weight gradients): the replicas still hold bit-identical parameters (every rank applies the same reduced
    gradient) and meet the same bound against the big batch."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), per_rank, deterministic), nprocs=world, join=True)
    ranks = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    r0 = ranks[0]
    for r in ranks[1:]:
        assert torch.equal(r0["params"], r["params"])  # replicas stay bit-identical: same reduced gradient, same Adam
    ns = _make(per_rank * world, 1, None, deterministic)
    for step in (1, 2):
        _run(ns, _batch(200 + step, per_rank * world), 0, per_rank * world)
    big = ns[0].params.cpu()
    nt = ns[0].trainable_numel
    delta = (big[:nt] - r0["params"][:nt]).abs().max().item()
    moved = (big[:nt] - torch.zeros(1)).abs().max().item()
    # 2 Adam steps of lr 1e-4 move weights by <= 2e-4; the two runs must agree to a small fraction of that
    assert delta <= 2.5e-4 and (big[:nt] - r0["params"][:nt]).abs().mean().item() < 2e-6
    # the per-rank partial losses sum to the big-batch loss
    assert abs(sum(r["loss"] for r in ranks).item() - ns[1].loss.item()) < 1e-5 * abs(ns[1].loss.item()) + 1e-7


# ---- ARCHITECTURE='basic': train-mode BatchNorm needs global statistics (SyncBN) for N ranks == one big batch ----
def _make_basic(B, world, hook=None, sync=None):
    from video_dqn_amd import synth
    from video_dqn_amd.engine import NetEngine, TDStepper
    net = NetEngine(3, 5, 1, False, "f32", 2 * B)
    net.load_tensors(synth.make_state_dict(7, extra_capacity=False))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, world_size=world, allreduce=hook)
    if sync is not None:
        net.set_bn_sync(world, sync)
    return net, stp


def _worker_basic(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)

    def through_host(t, stage=None):
        torch.cuda.synchronize()
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)

    ns = _make_basic(4, world, through_host, through_host)
    tup = _batch(231, 8)
    net, stp = ns
    dev = "cuda"
    lo, hi = rank * 4, rank * 4 + 4
    stp.forward_backward(tup[0][lo:hi].contiguous().to(dev), tup[1][lo:hi].contiguous().to(dev), 1, tup[2][lo:hi].to(dev),
                         tup[3][lo:hi].float().to(dev), tup[4][lo:hi].float().to(dev))
    torch.cuda.synchronize()
    torch.save({"grads": stp.grads.cpu(), "bnstats": net.bnstats.cpu(), "loss": stp.loss.cpu(), "q": stp.q_before.cpu()},
               os.path.join(out_dir, f"basic{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_basic_arch_syncbn_two_ranks_equal_one_big_batch(tmp_path):
    """With the SyncBN hook the BatchNorm statistics are global: the all-reduced gradient, the running statistics and
    Q(s) of two ranks x 4 samples equal one process with 8 samples (f32, up to summation order)."""
    world = 2
    mp.spawn(_worker_basic, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "basic0.pt")
    r1 = torch.load(tmp_path / "basic1.pt")
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["bnstats"], r1["bnstats"])
    net, stp = _make_basic(8, 1)
    tup = _batch(231, 8)
    dev = "cuda"
    stp.forward_backward(tup[0].contiguous().to(dev), tup[1].contiguous().to(dev), 1, tup[2].to(dev), tup[3].float().to(dev), tup[4].float().to(dev))
    torch.cuda.synchronize()
    big_g, big_s = stp.grads.cpu(), net.bnstats.cpu()

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert rel(r0["bnstats"], big_s) < 1e-5
    assert rel(torch.cat([r0["q"], r1["q"]]), stp.q_before.cpu()) < 1e-4
    assert abs((r0["loss"] + r1["loss"]).item() - stp.loss.item()) < 1e-4 * abs(stp.loss.item())
    # gradients: same network, same global statistics -> agreement to rounding (a ReLU flip would show as ~1e-2)
    assert rel(r0["grads"], big_g) < 2e-3
