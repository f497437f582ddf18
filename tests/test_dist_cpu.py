"""world_size-2 gloo test of the N>1 path: rank-local gradients computed with the GLOBAL loss normaliser and
summed by BucketAllReduce over the stage-ordered flat gradient equal the single-process big-batch gradient
(SURVEY.md §8e).  The per-rank arithmetic here is the CPU oracle (the HIP path needs a GPU); what is under
test is the product's exchange code: video_dqn_amd.dist + the engine's stage ranges / slot table."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_grads(model, target, cfg, tup, lo, hi, inv_count, ref_cpu):
    sub = tuple(t[lo:hi] for t in tup)
    d = {}
    model.zero_grad()
    ref_cpu.process_batch(model, target, cfg, sub, detail=d)
    # mean over the GLOBAL batch: sum of the local losses * inv_count  (what the fused TD kernel emits)
    loss = d["losses"].sum() * inv_count
    loss.backward()
    return loss.item()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ref_cpu
    from video_dqn_amd import synth
    from video_dqn_amd.dist import BucketAllReduce
    from video_dqn_amd.engine import NetEngine
    cfg = ref_cpu.default_config()
    B = 4
    (tup, _) = synth.make_batch(55, B, 1, structured=True, reward_p=0.3)
    tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7))
    tr.model.set_train()
    per = B // world
    loss = _rank_grads(tr.model, tr.target_net, cfg, tup, rank * per, (rank + 1) * per, 1.0 / (5 * B), ref_cpu)
    # flat gradient in the engine's stage order (storage-only engine: no GPU needed)
    eng = NetEngine(3, 5, 1, True, "f32", 8, device="cpu")
    flat = torch.zeros(eng.trainable_numel)
    named = dict(tr.model.named_parameters())
    for s in eng.slots.values():
        if s.kind == 0:
            flat[s.offset:s.offset + s.numel] = named[s.name].grad.reshape(-1)
    comm = BucketAllReduce(world)
    ranges = [eng.stage_range(st) for st in range(3)]
    assert ranges[0][0] == 0 and ranges[2][1] == eng.trainable_numel and ranges[0][1] == ranges[1][0] and ranges[1][1] == ranges[2][0]
    for st, (b, e) in enumerate(ranges):
        comm.launch(flat[b:e], st)
    comm.finish()
    lt = torch.tensor([loss])
    dist.all_reduce(lt)
    if rank == 0:
        torch.save({"flat": flat, "loss": lt.item()}, os.path.join(out_dir, "ddp.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_equals_big_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(tmp_path / "ddp.pt")
    sys.path.insert(0, ROOT)
    from oracle import ref_cpu
    from video_dqn_amd import synth
    from video_dqn_amd.engine import NetEngine
    cfg = ref_cpu.default_config()
    (tup, _) = synth.make_batch(55, 4, 1, structured=True, reward_p=0.3)
    tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7))
    tr.model.set_train()
    tr.model.zero_grad()
    loss = ref_cpu.process_batch(tr.model, tr.target_net, cfg, tup)  # .mean() over the whole batch
    loss.backward()
    eng = NetEngine(3, 5, 1, True, "f32", 8, device="cpu")
    assert abs(got["loss"] - loss.item()) < 1e-6 * abs(loss.item()) + 1e-9
    for s in eng.slots.values():
        if s.kind != 0:
            continue
        ref = dict(tr.model.named_parameters())[s.name].grad.reshape(-1)
        g = got["flat"][s.offset:s.offset + s.numel]
        assert (g - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-12, s.name


def _worker_loss(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from video_dqn_amd.dist import BucketAllReduce
    comm = BucketAllReduce(world)
    got = []
    for step in range(3):  # the two buffers are used in turn; a result is only valid until the launch after next
        comm.launch_loss(torch.tensor([float(rank + 1) * (step + 1)]))
        buf, work = comm.take_loss()
        work.wait()
        got.append(float(buf))
        assert comm.take_loss() is None  # taken once
    if rank == 0:
        torch.save(got, os.path.join(out_dir, "loss.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_loss_shares_are_summed_off_the_compute_stream(tmp_path):
    """BucketAllReduce.launch_loss / take_loss (the trainer's running loss under N > 1: every rank's scalar is its share of the
    global batch mean, train_q_network.py:180,228-231): the SUM over ranks arrives in a buffer of its own, once per launch."""
    world = 2
    mp.spawn(_worker_loss, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert torch.load(tmp_path / "loss.pt") == [3.0, 6.0, 9.0]
    from video_dqn_amd.dist import BucketAllReduce
    single = BucketAllReduce(1)
    single.launch_loss(torch.tensor([1.0]))  # one process: nothing to reduce, nothing pending
    assert single.take_loss() is None
