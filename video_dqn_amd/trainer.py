"""The Q-learning training loop with the reference's control flow and on-disk artefacts
(``train_q_network.py:84-250``), running every device-side step on the HIP engine.

Kept from the reference: seeding (:86-87), dataset/DataLoader construction (:98-114; batch size is a config
key here), model + target + Adam (:119-124), resume (:192-198), target sync before the update when
``sample_number % TARGET_UPDATE_INTERVAL == 0`` (:215-216), EMA of the loss and its tensorboard scalar
(:228-238), checkpoint dict and file name (:241-247).  Not kept: the crash after the first checkpoint
(:248-250) and the per-step host sync of ``loss.item()`` — the loss is read back one step late
from pinned memory so the GPU never idles.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch
from torch.utils import data

from .dataset import QLearningRealDataset, SyntheticTupleDataset
from .dist import BucketAllReduce, agree_all, broadcast_replica_state
from .engine import TDStepper
from .model import build_model
from .shards import ShardDataset, is_shard_dir


def loopLoader(loader, on_reset=None):
    """train_q_network.py:60-67."""
    i = iter(loader)
    epoch = 0
    while True:
        try:
            yield next(i)
        except StopIteration:
            print("reset iterator")
            epoch += 1
            if on_reset:
                on_reset(epoch)
            i = iter(loader)


# ---- checkpoint: exactly the reference's dict (train_q_network.py:241-247) --------------------------------
def optimizer_state_dict(stepper: TDStepper) -> dict:
    """torch.optim.Adam.state_dict() layout: ids follow model.parameters() order (70 ids; resnet.fc = 60, 61
    never receives a gradient, so it has no state — as in the reference)."""
    net = stepper.net
    state = {}
    if stepper.adam_step > 0:
        for s in net.slots.values():
            if s.kind != 0:
                continue
            sl = slice(s.offset, s.offset + s.numel)
            state[s.param_id] = {"step": stepper.adam_step,
                                 "exp_avg": stepper.exp_avg[sl].view(s.shape).clone(),
                                 "exp_avg_sq": stepper.exp_avg_sq[sl].view(s.shape).clone()}
    n_params = sum(1 for s in net.slots.values() if s.kind in (0, 1))
    group = {"lr": stepper.lr, "betas": tuple(stepper.betas), "eps": stepper.eps, "weight_decay": 0, "amsgrad": False,
             "params": list(range(n_params))}
    return {"state": dict(sorted(state.items())), "param_groups": [group]}


def load_optimizer_state_dict(stepper: TDStepper, sd: dict) -> None:
    net = stepper.net
    by_id = {s.param_id: s for s in net.slots.values() if s.kind == 0}
    step = 0
    with torch.no_grad():
        for pid, st in sd["state"].items():
            s = by_id[int(pid)]
            sl = slice(s.offset, s.offset + s.numel)
            stepper.exp_avg[sl].copy_(st["exp_avg"].reshape(-1))
            stepper.exp_avg_sq[sl].copy_(st["exp_avg_sq"].reshape(-1))
            step = int(st["step"])
    stepper.adam_step = step
    g = sd["param_groups"][0]
    stepper.lr, stepper.betas, stepper.eps = g["lr"], tuple(g["betas"]), g["eps"]


def save_checkpoint(path, sample_number, model, stepper):
    torch.save({"sample_number": sample_number,
                "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer_state_dict(stepper)}, path)


def _to_device_batch(batch, device, num_classes=5):
    before, after, act, rew, term, gt, valid = batch
    nb = dict(non_blocking=True)
    before, after = before.to(device, **nb), after.to(device, **nb)
    src_kind = 0 if before.dtype == torch.uint8 else 1
    if src_kind == 1:
        before, after = before.float().contiguous(), after.float().contiguous()
    act = torch.as_tensor(act).to(torch.int64).to(device, **nb)
    rew = torch.as_tensor(rew).float().to(device, **nb)
    term = torch.as_tensor(term).float().to(device, **nb)
    valid = torch.as_tensor(valid).float().to(device, **nb)
    gt = torch.as_tensor(gt).float()
    if gt.dim() == 1:
        gt = gt.view(-1, 1).expand(-1, num_classes)
    gt = gt.contiguous().to(device, **nb)
    return before.contiguous(), after.contiguous(), src_kind, act, rew, term, valid, gt


class DevicePrefetcher:
    """Keeps one batch ahead on the device: the pinned host batch of step t+1 is copied on a separate stream while the
    kernels of step t run (measured with bench.py --h2d: copies on the compute stream cost 2.2 ms per 256-sample step,
    double-buffered on a copy stream they cost nothing)."""

    def __init__(self, iterator, device):
        self.it, self.device = iterator, device
        self.stream = torch.cuda.Stream(device=device)
        self._next = None
        self._fill()

    def _fill(self):
        batch = next(self.it)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))  # the buffers of two steps ago are free again
        with torch.cuda.stream(self.stream):
            dev_batch = _to_device_batch(batch, self.device)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._next = (dev_batch, ev)

    def __next__(self):
        dev_batch, ev = self._next
        torch.cuda.current_stream(self.device).wait_event(ev)
        for t in dev_batch:
            if torch.is_tensor(t):
                t.record_stream(torch.cuda.current_stream(self.device))  # allocated on the copy stream, consumed on the compute stream
        self._fill()
        return dev_batch


def run_train(config, resume_from=-1, max_steps=None, rank=0, world_size=1, log=print):
    """train_q_network.py:84-250."""
    torch.manual_seed(config.SEED)
    np.random.seed(config.SEED)
    log(f"Using: {config.device}")
    B = int(config.BATCH_SIZE)
    params = {"batch_size": B, "num_workers": int(config.NUM_WORKERS), "drop_last": True}
    store = None  # DeviceFrameStore when the decoded frames live in HBM
    stream = None  # HostFrameStream when they are streamed from memory-mapped shards
    if config.SYNTHETIC_DATA or config.DATASET in ("none", "synthetic"):
        nf = config.NUM_FRAMES or (4 if (config.PANORAMA or config.PREVIOUS_IMAGES) else 1)
        dataset = SyntheticTupleDataset(length=max(4 * B * world_size, 1024), num_frames=nf,
                                        action_dim=1 if (config.VALUE_LEARNING or config.ONE_ACTION) else 3, seed=config.SEED)
    else:
        kw = dict(one_action=True, confidence_reward=config.CONFIDENCE_REWARD, value_learning=config.VALUE_LEARNING,
                  inverse_actions=config.USE_INVERSE_ACTIONS, previous_images=config.PREVIOUS_IMAGES)
        if is_shard_dir(config.DATASET):  # pre-decoded uint8 frames (python -m video_dqn_amd.shards)
            dataset = ShardDataset(config.DATASET, **kw)
            resident = str(getattr(config, "DEVICE_RESIDENT_DATA", "auto")).lower()
            if resident not in ("auto", "on", "off"):
                raise ValueError("DEVICE_RESIDENT_DATA must be 'auto', 'on' or 'off'")
            sharded = world_size > 1 and bool(getattr(config, "RANK_SHARDED_DATA", True))
            gather_threads = int(getattr(config, "HOST_GATHER_THREADS", 0))
            headroom = 32 << 30  # activations, workspaces, allocator slack
            if sharded and resident != "off":
                # N ranks: each holds only the frames its samples of the current epoch reference (re-uploaded per epoch), not a full
                # copy of the dataset; same minibatch sequence as the full per-rank copy.  Residency is decided from THIS rank's
                # subset (epoch 0's distinct frames + 10 % for later epochs), one decision for the whole job: a rank on another
                # input path would draw from a different sharding of the epoch than the resident ones
                from .shards import RankShardedFrameStore
                store = RankShardedFrameStore(config.DATASET, config.device, rank, world_size, threads=gather_threads, **kw)
                need = store.epoch_frames(B, config.SEED) * 224 * 224 * 3
                free, _ = torch.cuda.mem_get_info(torch.device(config.device))
                fits = agree_all(need + need // 10 + headroom < free, device=config.device)
                if not fits and resident == "on":
                    raise RuntimeError(f"DEVICE_RESIDENT_DATA: 'on', but this rank's share of the frames ({need / 2**30:.1f} GiB + "
                                       f"{headroom >> 30} GiB of headroom) does not fit in {free / 2**30:.1f} GiB of free HBM on every rank")
                if fits:
                    log(f"dataset resident in HBM, sharded by rank: the frames of this rank's samples are uploaded per epoch "
                        f"({need / 2**30:.2f} GiB of the dataset's {store.total_frames * 224 * 224 * 3 / 2**30:.2f} GiB)")
                else:
                    store = None
                    log(f"this rank's share of the frames ({need / 2**30:.1f} GiB) does not fit in HBM on every rank: streaming")
                resident = fits
            elif resident != "auto":
                resident = resident == "on"
            else:  # frames + 32 GiB of headroom must fit in the GPU's free memory
                n_frames = sum(np.load(p_, mmap_mode="r").shape[0] for p_ in dataset._paths)
                free, _ = torch.cuda.mem_get_info(torch.device(config.device))
                resident = n_frames * 224 * 224 * 3 + headroom < free
                if world_size > 1:
                    resident = agree_all(resident, device=config.device)
            if store is not None:
                pass
            elif resident:
                from .shards import DeviceFrameStore
                store = DeviceFrameStore(config.DATASET, config.device, **kw)
                log(f"dataset resident in HBM: {store.bytes() / 2**30:.2f} GiB of frames")
            elif str(getattr(config, "SHARD_INPUT", "stream")).lower() == "stream":
                # the frames do not fit (or residency is off): memory-mapped shards, native gather into pinned double buffers,
                # host-to-device copies on a prefetch stream — same minibatch sequence as the resident store
                from .shards import HostFrameStream
                stream = HostFrameStream(config.DATASET, config.device, B, config.SEED, rank, world_size,
                                         threads=gather_threads, **kw)
                log(f"dataset streamed from memory-mapped shards: {stream.threads} gather threads, {len(stream._slots)} pinned staging buffers")
        else:
            dataset = QLearningRealDataset(config.DATASET, as_uint8=True, **kw)
        log(f"Load data from {config.DATASET}")
        log(f"Reward Ratio: {dataset.reward_percentage()}")
    sampler = None
    if world_size > 1:
        sampler = data.distributed.DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=True,
                                                      seed=config.SEED, drop_last=True)
    extra = {}
    if isinstance(dataset, ShardDataset):  # batched fetch: one gather per shard and batch instead of per-sample copies + collate
        from .shards import collate_batches
        dataset.batched_fetch = True
        extra["collate_fn"] = collate_batches
    loader = data.DataLoader(dataset, **params, shuffle=(sampler is None), sampler=sampler, pin_memory=True,
                             persistent_workers=params["num_workers"] > 0, **extra)
    log(len(dataset))

    model = build_model(config, max_batch=2 * B)
    comm = BucketAllReduce(world_size) if world_size > 1 else None
    stepper = TDStepper(model.engine, B, lr=config.LEARNING_RATE, gamma=config.GAMMA,
                        clip_rect=(config.LOSS_CLIP == "rect"), linear=config.LINEAR,
                        remove_before_reward=config.REMOVE_BEFORE_REWARD,
                        train_on_ground_truth=config.TRAIN_ON_GROUND_TRUTH, value_learning=config.VALUE_LEARNING,
                        target_update_interval=config.TARGET_UPDATE_INTERVAL, world_size=world_size,
                        allreduce=(comm.launch if comm else None), loss_kind=getattr(config, "LOSS_KIND", "l2"),
                        allreduce_loss=(comm.launch_loss if comm else None), allreduce_wait=(comm.wait_last if comm else None))
    if world_size > 1 and config.ARCHITECTURE != "extra_capacity" and getattr(config, "SYNC_BN", True):
        model.engine.set_bn_sync(world_size)  # train-mode BatchNorm over the global batch, as the single-GPU reference sees it
    if store is not None:  # minibatches are gathered on the device; no loader, no host copies
        from .shards import RankShardedFrameStore as _RS
        iterator = store.batches(B, config.SEED) if isinstance(store, _RS) else store.batches(B, config.SEED, rank, world_size)
    elif stream is not None:
        iterator = stream.batches()
    else:
        iterator = DevicePrefetcher(loopLoader(loader, on_reset=(sampler.set_epoch if sampler else None)), model.engine.device)
    os.makedirs(f"{config.folder}/models", exist_ok=True)
    sample_number = resume_from + 1
    if resume_from > -1:  # :192-198
        model_loc = f"{config.folder}/models/sample{resume_from}.torch"
        snapshot = torch.load(model_loc, map_location=config.device)
        log(f"Loading model from: {model_loc}")
        model.load_state_dict(snapshot["model_state_dict"])
        load_optimizer_state_dict(stepper, snapshot["optimizer_state_dict"])
    if config.BOOTSTRAP:  # :200-206 — start from a network trained on ground truth (model AND optimiser state)
        log("\n\nBOOTSTRAP\n\n")
        model_loc = getattr(config, "BOOTSTRAP_CHECKPOINT", "") or "logs/trained_gt_0.99/models/epoch99.torch"  # :202
        snapshot = torch.load(model_loc, map_location=config.device)  # a missing file raises, as in the reference
        log(f"Loading model from: {model_loc}")
        model.load_state_dict(snapshot["model_state_dict"])
        load_optimizer_state_dict(stepper, snapshot["optimizer_state_dict"])
    if resume_from < 0 and not config.BOOTSTRAP and not getattr(model, "pretrained_loaded", False) and rank == 0:
        log("WARNING: the ResNet-18 trunk starts from a RANDOM initialisation — the reference builds "
            "models.resnet18(pretrained=True) (archs/HabitatDQNMultiAction.py:11); set PRETRAINED_WEIGHTS in config.yml "
            "(a torchvision resnet18 state_dict file) to train on ImageNet features as the reference does")
    if world_size > 1:  # replicas must start identical: rank 0's parameters, statistics and optimiser state everywhere
        eng = model.engine
        broadcast_replica_state([eng.params, eng.bnstats, eng.num_batches_tracked, stepper.exp_avg, stepper.exp_avg_sq])
        steps = torch.tensor([stepper.adam_step], dtype=torch.int64, device=eng.device)
        torch.distributed.broadcast(steps, src=0)
        stepper.adam_step = int(steps.item())
        eng.mark_dirty()
    stepper.sync_target()  # :208
    stepper.sample_number = sample_number

    running_loss = None
    host_loss = torch.zeros(2, dtype=torch.float32).pin_memory()
    pending = None  # (slot, event) of the previous step's loss copy
    loss_stream = None  # N > 1: where the all-reduced loss is waited for and copied to the host
    num_steps = config.NUM_STEPS if max_steps is None else min(config.NUM_STEPS, sample_number + max_steps)

    def consume(p):
        nonlocal running_loss
        slot, ev = p
        ev.synchronize()
        v = float(host_loss[slot])
        running_loss = v if running_loss is None else running_loss * 0.99 + v * 0.01  # :228-231

    try:  # (the streaming input path owns a thread, pinned buffers and a prefetch stream: released on every exit)
        while sample_number < num_steps:
            sample_number += 1
            model.set_train()  # :221 (flags only; the engine's BatchNorm is always in eval mode in extra_capacity)
            before, after, src_kind, act, rew, term, valid, gt = next(iterator)
            # the stepper performs the :215-216 target refresh itself (sample_number % TARGET_UPDATE_INTERVAL == 0)
            loss = stepper.step(before, after, src_kind, act, rew, term,
                                valid if config.REMOVE_BEFORE_REWARD else None,
                                gt if config.TRAIN_ON_GROUND_TRUTH else None,
                                finish_allreduce=(comm.finish if comm else None))
            # every rank's `loss` is its share of the global mean (the TD kernel divides by the global batch): their SUM is the
            # batch-mean loss the reference feeds into its running average every update (:228-231).  The stepper has queued that
            # 4-byte all-reduce on the gradient stream behind the last gradient bucket (dist.launch_loss); it is waited for and
            # copied to the host on the read-back stream only — the compute stream, i.e. the next update's first kernel, never
            # waits for it — so the average that is printed, logged and returned is still the reference's: an EMA of the true
            # global batch-mean loss, one update late like the single-process read-back
            slot = sample_number & 1
            ev = torch.cuda.Event()
            reduced = comm.take_loss() if comm is not None else None
            if reduced is not None:
                buf, work = reduced
                if loss_stream is None:
                    loss_stream = torch.cuda.Stream(device=model.engine.device)
                with torch.cuda.stream(loss_stream):
                    work.wait()
                    host_loss[slot:slot + 1].copy_(buf, non_blocking=True)
                    ev.record(loss_stream)
            else:
                host_loss[slot:slot + 1].copy_(loss, non_blocking=True)
                ev.record()
            if pending is not None:
                consume(pending)
            pending = (slot, ev)
            log_now = sample_number % 100 == 0 and rank == 0 and hasattr(config, "writer")
            if log_now:  # the reference logs the average INCLUDING this update's loss (:228-238): take it in before writing
                consume(pending)
                pending = None
            if rank == 0 and running_loss is not None:
                print(f"\rbatch:{sample_number}/{config.NUM_STEPS} avg_loss: {running_loss}", end="")
            if log_now and running_loss is not None:
                config.writer.add_scalar("avg_q_loss/train", running_loss, sample_number)  # :236-238
            if sample_number % config.CHECKPOINT_INTERVAL == 0 and rank == 0:  # :241-247
                torch.cuda.synchronize()
                save_checkpoint(f"{config.folder}/models/sample{sample_number}.torch", sample_number, model, stepper)
        if pending is not None:
            consume(pending)
        torch.cuda.synchronize()
    finally:
        if stream is not None:
            stream.close()
    return model, stepper, running_loss
