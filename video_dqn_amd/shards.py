"""Decoded-frame shards: the input side of the hot path without per-step JPEG decoding.

On real data the reference's loader (``dataloaders/q_learning_real.py:55-73``) decodes + resizes two (or eight,
with PREVIOUS_IMAGES) JPEGs per sample in DataLoader workers — a few hundred frames/s, two orders of magnitude below
what the GPU path consumes.  ``build_shards`` runs that transform ONCE: every distinct frame a feather file references
is decoded with exactly the loader's ``Resize(224) + CenterCrop(224)`` (bit-identical: the same function) and stored as
raw uint8 224x224x3 in ``frames_<k>.u8``; an ``index.npz`` sidecar holds, per sample, the frame indices (incl. the
clamped ``id, id-1, id-2, id-3`` gather of ``:57-70`` as precomputed index arithmetic) and the label columns of the
feather schema (``dataset/process_episodes_real.py:144-181``).  ``ShardDataset`` memory-maps the shards and returns the
same 7-tuple as ``QLearningRealDataset(as_uint8=True)``; normalisation happens in the GPU input-packing kernel.

    python -m video_dqn_amd.shards dataset/data.feather dataset/shards     # one-time conversion
    DATASET: 'dataset/shards'                                               # in config.yml
"""
from __future__ import annotations

import os
import re
import sys

import numpy as np
import torch
from torch.utils import data

from .dataset import detection_thresholds, multi_get, resize_center_crop_u8

FRAME_BYTES = 224 * 224 * 3


def _prev_paths(path: str, start: int):
    """frames id, id-1, id-2, id-3 clamped at the episode start (dataloaders/q_learning_real.py:60-67)."""
    m = re.match(r"(.*?/)(\d+).jpg", path)
    prefix, im_id = m[1], int(m[2])
    return [prefix + "%04d.jpg" % max(im_id - i, start) for i in range(4)]


def build_shards(feather_path: str, out_dir: str, shard_frames: int = 2048, with_previous: bool = True, log=print) -> dict:
    import pandas as pd
    from PIL import Image
    df = pd.read_feather(feather_path)
    n = len(df)
    os.makedirs(out_dir, exist_ok=True)
    frame_id = {}
    order = []

    def fid(path):
        i = frame_id.get(path)
        if i is None:
            i = frame_id[path] = len(order)
            order.append(path)
        return i

    before = np.empty((n, 4), dtype=np.int64)
    after = np.empty((n, 4), dtype=np.int64)
    for r, (bp, ap, start) in enumerate(zip(df["before_image"], df["after_image"], df["im_start"])):
        if with_previous:
            before[r] = [fid(p) for p in _prev_paths(bp, int(start))]
            after[r] = [fid(p) for p in _prev_paths(ap, int(start))]
        else:
            before[r] = fid(bp)
            after[r] = fid(ap)
    n_frames = len(order)
    n_shards = (n_frames + shard_frames - 1) // shard_frames
    for s in range(n_shards):
        lo, hi = s * shard_frames, min(n_frames, (s + 1) * shard_frames)
        mm = np.lib.format.open_memmap(os.path.join(out_dir, f"frames_{s:05d}.npy"), mode="w+", dtype=np.uint8, shape=(hi - lo, 224, 224, 3))
        for i in range(lo, hi):
            mm[i - lo] = resize_center_crop_u8(Image.open(order[i]))
        mm.flush()
        del mm
        log(f"shard {s + 1}/{n_shards}: frames {lo}..{hi - 1}")
    cols = {"before": before, "after": after, "shard_frames": np.int64(shard_frames), "n_frames": np.int64(n_frames),
            "detector_score": multi_get(df, "detector_score").astype(np.float64),
            "sparse_reward": multi_get(df, "sparse_reward").astype(np.int64),
            "steps_to_reward": multi_get(df, "steps_to_reward").astype(np.float64),
            "inverse_actions": (df["inverse_actions"].to_numpy().astype(np.int64) if "inverse_actions" in df else np.zeros(n, np.int64)),
            "has_inverse_actions": np.int64("inverse_actions" in df), "with_previous": np.int64(with_previous)}
    np.savez(os.path.join(out_dir, "index.npz"), **cols)
    return {"samples": n, "frames": n_frames, "shards": n_shards, "bytes": n_frames * FRAME_BYTES}


def is_shard_dir(path: str) -> bool:
    return os.path.isdir(path) and os.path.exists(os.path.join(path, "index.npz"))


class ShardDataset(data.Dataset):
    """Same constructor flags and 7-tuple as QLearningRealDataset (frames as uint8 HWC, or [4,H,W,3] stacks)."""

    def __init__(self, location, one_action=False, value_learning=False, inverse_actions=False, previous_images=False,
                 confidence_reward=False, slam_actions=False, gamma=0.99):
        idx = np.load(os.path.join(location, "index.npz"))
        self.before, self.after = idx["before"], idx["after"]
        self.detector_score, self.sparse_reward = idx["detector_score"], idx["sparse_reward"]
        self.steps_to_reward, self.actions = idx["steps_to_reward"], idx["inverse_actions"]
        self.shard_frames = int(idx["shard_frames"])
        n_frames = int(idx["n_frames"])
        if previous_images and not int(idx["with_previous"]):
            raise ValueError("these shards were built without the PREVIOUS_IMAGES frame gather")
        if inverse_actions and not int(idx["has_inverse_actions"]):
            raise KeyError("inverse_actions")
        n_shards = (n_frames + self.shard_frames - 1) // self.shard_frames
        self._paths = [os.path.join(location, f"frames_{s:05d}.npy") for s in range(n_shards)]
        self._maps = None  # opened lazily (per DataLoader worker)
        self.value_learning, self.confidence_reward, self.slam_actions = value_learning, confidence_reward, slam_actions
        self.one_action, self.inverse_actions, self.gamma, self.previous_images = one_action, inverse_actions, gamma, previous_images

    def __len__(self):
        return len(self.before)

    def reward_percentage(self):  # dataloaders/q_learning_real.py:50-52
        return (self.sparse_reward.max(axis=1) > 0).sum() / self.sparse_reward.shape[0]

    def _frame(self, i):
        if self._maps is None:
            self._maps = [np.load(p, mmap_mode="r") for p in self._paths]
        return torch.from_numpy(np.array(self._maps[i // self.shard_frames][i % self.shard_frames]))

    def __getitems__(self, indices):
        """Batched fetch (torch DataLoader calls this with the whole index list of a batch when it exists): frames are
        gathered shard by shard with one fancy-indexed copy each, straight into the batch tensors — no per-sample Python
        work, no second collate copy.  Opt-in: set ``dataset.batched_fetch = True`` AND pass ``collate_fn=collate_batches``
        (then the already collated 7-tuple is returned); otherwise this behaves like per-sample ``__getitem__``."""
        if not getattr(self, "batched_fetch", False):  # plain DataLoader (default collate): per-sample semantics
            return [self[int(i)] for i in indices]
        idx = np.asarray(indices, dtype=np.int64)
        if self._maps is None:
            self._maps = [np.load(p, mmap_mode="r") for p in self._paths]
        nf = 4 if self.previous_images else 1

        def gather(table):
            fr = table[idx][:, :nf].reshape(-1)
            out = np.empty((fr.shape[0], 224, 224, 3), dtype=np.uint8)
            shard = fr // self.shard_frames
            for s in np.unique(shard):
                sel = np.nonzero(shard == s)[0]
                loc = fr[sel] % self.shard_frames
                order = np.argsort(loc, kind="stable")
                out[sel[order]] = self._maps[s][loc[order]]
            t = torch.from_numpy(out)
            return t.view(len(idx), nf, 224, 224, 3) if self.previous_images else t
        bi, ai = gather(self.before), gather(self.after)
        detections = self.detector_score[idx]
        reward = detections if self.confidence_reward else (detections > detection_thresholds).astype(np.int64)
        valid = np.ones_like(reward)
        if self.value_learning:
            steps = self.steps_to_reward[idx]
            gt = np.power(np.ones((len(idx), 5)) * self.gamma, steps)
            gt[steps == np.inf] = np.nan
        else:
            gt = np.full((len(idx),), np.nan)
        if self.inverse_actions:
            action = self.actions[idx].astype(np.int64)
        elif self.slam_actions:
            raise NotImplementedError("not implemented")
        elif self.one_action:
            action = np.zeros(len(idx), dtype=np.int64)
        else:
            raise Exception("not implemented")
        rew_t = torch.from_numpy(np.ascontiguousarray(reward))
        return [(bi, ai, torch.from_numpy(action), rew_t, rew_t.clone(), torch.from_numpy(gt), torch.from_numpy(valid))]

    def __getitem__(self, index):
        if self.previous_images:
            bi = torch.stack([self._frame(int(i)) for i in self.before[index]])
            ai = torch.stack([self._frame(int(i)) for i in self.after[index]])
        else:
            bi, ai = self._frame(int(self.before[index, 0])), self._frame(int(self.after[index, 0]))
        detections = self.detector_score[index]
        if self.confidence_reward:
            reward = detections
        else:
            reward = (detections > detection_thresholds).astype(np.int64)
        valid_mask = np.ones_like(reward)
        gt = np.nan
        if self.value_learning:
            steps = self.steps_to_reward[index]
            gt = np.power(np.ones((5,)) * self.gamma, steps)
            gt[steps == np.inf] = np.nan
        if self.inverse_actions:
            action = int(self.actions[index])
        elif self.slam_actions:
            raise NotImplementedError("not implemented")
        elif self.one_action:
            action = 0
        else:
            raise Exception("not implemented")
        return bi, ai, action, reward, reward, gt, valid_mask


class DeviceFrameStore:
    """The whole decoded-frame dataset resident in HBM (MI355X: 288 GB = 1.9 M frames of 224x224x3 uint8).

    The DataLoader path tops out at 10-20 k samples/s on the GPU box's host (tools/bench_loader.py: worker -> shared
    memory -> pinned memory copies of 77 MB batches), a third to a half of what one GPU consumes (34 k tuples/s).  With
    the frames in HBM a minibatch is an index gather on the device — no host traffic at all — and the loop runs at the
    speed of bench.py.  Semantics are those of ``ShardDataset`` (same labels, same 7-tuple, frames as uint8) with the
    loader's shuffle + drop_last (train_q_network.py:98,114) done by a seeded device permutation per epoch; under data
    parallelism every rank holds the full store and takes the rank-strided slice of each epoch's permutation."""

    def __init__(self, location, device, **flags):
        ds = ShardDataset(location, **flags)
        self.flags = ds
        self.device = torch.device(device)
        maps = [np.load(p, mmap_mode="r") for p in ds._paths]
        n_frames = sum(m.shape[0] for m in maps)
        self.frames = torch.empty((n_frames, 224, 224, 3), dtype=torch.uint8, device=self.device)
        lo = 0
        for m in maps:  # shard by shard through pinned staging: never more than one shard on the host
            t = torch.from_numpy(np.ascontiguousarray(m))
            self.frames[lo:lo + t.shape[0]].copy_(t.pin_memory() if torch.cuda.is_available() else t, non_blocking=False)
            lo += t.shape[0]
        nf = 4 if ds.previous_images else 1
        self.nf = nf
        self.before = torch.from_numpy(ds.before[:, :nf].astype(np.int64)).to(self.device)
        self.after = torch.from_numpy(ds.after[:, :nf].astype(np.int64)).to(self.device)
        all_idx = np.arange(len(ds))
        lab = self._labels(ds, all_idx)
        self.act, self.rew, self.term, self.gt, self.valid = (x.to(self.device) for x in lab)

    @staticmethod
    def _labels(ds, idx):
        detections = ds.detector_score[idx]
        reward = detections if ds.confidence_reward else (detections > detection_thresholds).astype(np.int64)
        valid = np.ones_like(reward)
        if ds.value_learning:
            steps = ds.steps_to_reward[idx]
            gt = np.power(np.ones((len(idx), 5)) * ds.gamma, steps)
            gt[steps == np.inf] = np.nan
        else:
            gt = np.full((len(idx), 5), np.nan)
        if ds.inverse_actions:
            action = ds.actions[idx].astype(np.int64)
        elif ds.slam_actions:
            raise NotImplementedError("not implemented")
        elif ds.one_action:
            action = np.zeros(len(idx), dtype=np.int64)
        else:
            raise Exception("not implemented")
        rew_t = torch.from_numpy(np.ascontiguousarray(reward)).float()
        return (torch.from_numpy(action), rew_t, rew_t.clone(), torch.from_numpy(gt).float(), torch.from_numpy(valid).float())

    def __len__(self):
        return self.before.shape[0]

    def bytes(self):
        return self.frames.numel()

    def gather(self, idx: torch.Tensor):
        """Device batch for sample indices ``idx`` (int64 device tensor), in the order trainer._to_device_batch produces:
        (before, after, src_kind=0, act, rew, term, valid, gt)."""
        b = self.frames.index_select(0, self.before.index_select(0, idx).reshape(-1))
        a = self.frames.index_select(0, self.after.index_select(0, idx).reshape(-1))
        if self.nf > 1:
            b = b.view(idx.shape[0], self.nf, 224, 224, 3)
            a = a.view(idx.shape[0], self.nf, 224, 224, 3)
        return (b, a, 0, self.act.index_select(0, idx), self.rew.index_select(0, idx), self.term.index_select(0, idx),
                self.valid.index_select(0, idx), self.gt.index_select(0, idx))

    def batches(self, batch_size: int, seed: int, rank: int = 0, world_size: int = 1):
        """Endless stream of device batches: per epoch one seeded permutation (identical on every rank), rank r takes
        perm[r::world], drop_last."""
        epoch = 0
        n = len(self)
        per_rank = (n // world_size // batch_size) * batch_size
        if per_rank == 0:
            raise ValueError(f"dataset of {n} samples is smaller than one global batch ({batch_size} x {world_size})")
        while True:
            g = torch.Generator(device="cpu")
            g.manual_seed(seed + epoch)
            perm = torch.randperm(n, generator=g)[rank::world_size][:per_rank].to(self.device)
            for lo in range(0, per_rank, batch_size):
                yield self.gather(perm[lo:lo + batch_size])
            epoch += 1
            print("reset iterator")


class RankShardedFrameStore:
    """HBM-resident frames under data parallelism WITHOUT a full copy per rank (round-4 review: eight ranks each holding the whole
    dataset).  Rank r draws the samples perm_e[r::world] of epoch e (the loader's DistributedSampler split, train_q_network.py:98,114 on
    one GPU); at the start of every epoch it uploads exactly the frames those samples reference — gathered from the memory-mapped
    shards by ``vdqn_host_gather`` into a pinned staging buffer, chunk by chunk — and gathers its minibatches from that resident
    subset.  The minibatch sequence is bit for bit that of ``DeviceFrameStore.batches(B, seed, rank, world)``.  Memory per rank: the
    distinct frames referenced by 1/world of the samples — 1/world of the dataset when samples do not share frames, more when
    they do (with PREVIOUS_IMAGES every sample references eight frames of its episode: a random quarter of the samples touches most
    frames; the trainer then falls back to the streaming path if the subset does not fit)."""

    def __init__(self, location, device, rank: int, world_size: int, threads: int = 0, chunk_frames: int = 1024, **flags):
        from . import _lib
        ds = ShardDataset(location, **flags)
        self.flags = ds
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.rank, self.world = int(rank), int(world_size)
        self.nf = 4 if ds.previous_images else 1
        self.n = len(ds)
        self._maps = [np.load(p, mmap_mode="r") for p in ds._paths]
        self._base = np.array([m.ctypes.data for m in self._maps], dtype=np.uint64)
        self._shard_frames = ds.shard_frames
        self._before = np.ascontiguousarray(ds.before[:, :self.nf].astype(np.int64))
        self._after = np.ascontiguousarray(ds.after[:, :self.nf].astype(np.int64))
        lab = DeviceFrameStore._labels(ds, np.arange(self.n))
        self.act, self.rew, self.term, self.gt, self.valid = (x.to(self.device) for x in lab)
        self.threads = int(threads) if threads else max(1, min(16, (os.cpu_count() or 2) // 2))
        self._lib = _lib.load()
        self._chunk = int(chunk_frames)
        self._stage = []
        for _ in range(2):
            stage = torch.empty((self._chunk, 224, 224, 3), dtype=torch.uint8)
            self._stage.append(stage.pin_memory() if self.cuda else stage)
        self.frames = None
        self.resident_frames = 0
        self.total_frames = sum(m.shape[0] for m in self._maps)

    def __len__(self):
        return self.n

    def bytes(self):
        """Bytes of frames resident on this rank for the current epoch."""
        return self.resident_frames * FRAME_BYTES

    def _epoch_samples(self, batch_size: int, seed: int, epoch: int) -> np.ndarray:
        per_rank = (self.n // self.world // batch_size) * batch_size
        g = torch.Generator(device="cpu")
        g.manual_seed(seed + epoch)
        return torch.randperm(self.n, generator=g)[self.rank::self.world][:per_rank].numpy()

    def epoch_frames(self, batch_size: int, seed: int, epoch: int = 0) -> int:
        """Number of distinct frames this rank's samples of ``epoch`` reference (what _load_epoch will make resident)."""
        idx = self._epoch_samples(batch_size, seed, epoch)
        return int(np.unique(np.concatenate([self._before[idx].reshape(-1), self._after[idx].reshape(-1)])).shape[0])

    def _load_epoch(self, idx: np.ndarray):
        fb, fa = self._before[idx], self._after[idx]
        need = np.unique(np.concatenate([fb.reshape(-1), fa.reshape(-1)]))
        if self.frames is None or self.frames.shape[0] < need.shape[0]:
            self.frames = None  # release the old buffer first: the peak is one subset, not two
            cap = min(self.total_frames, need.shape[0] + need.shape[0] // 16 + 1)
            self.frames = torch.empty((cap, 224, 224, 3), dtype=torch.uint8, device=self.device)
        addr = self._base[need // self._shard_frames] + (need % self._shard_frames).astype(np.uint64) * np.uint64(FRAME_BYTES)
        busy = [None, None]  # event behind the last copy that read each staging buffer
        for k, lo in enumerate(range(0, need.shape[0], self._chunk)):
            a = np.ascontiguousarray(addr[lo:lo + self._chunk])
            stage = self._stage[k & 1]
            if busy[k & 1] is not None:
                busy[k & 1].synchronize()
            if self._lib.vdqn_host_gather(stage.data_ptr(), a.ctypes.data, a.shape[0], FRAME_BYTES, self.threads) != 0:
                raise RuntimeError("vdqn_host_gather failed")
            self.frames[lo:lo + a.shape[0]].copy_(stage[:a.shape[0]], non_blocking=self.cuda)
            if self.cuda:
                busy[k & 1] = torch.cuda.Event()
                busy[k & 1].record()
        for ev in busy:
            if ev is not None:
                ev.synchronize()
        self.resident_frames = int(need.shape[0])
        self._b_local = torch.from_numpy(np.searchsorted(need, fb)).to(self.device)  # [per_rank, nf] positions inside the resident subset
        self._a_local = torch.from_numpy(np.searchsorted(need, fa)).to(self.device)
        self._idx_dev = torch.from_numpy(idx).to(self.device)

    def batches(self, batch_size: int, seed: int):
        """Endless stream of device batches — DeviceFrameStore.batches(batch_size, seed, rank, world)'s sequence."""
        per_rank = (self.n // self.world // batch_size) * batch_size
        if per_rank == 0:
            raise ValueError(f"dataset of {self.n} samples is smaller than one global batch ({batch_size} x {self.world})")
        epoch = 0
        while True:
            idx = self._epoch_samples(batch_size, seed, epoch)
            self._load_epoch(idx)
            for lo in range(0, per_rank, batch_size):
                sl = slice(lo, lo + batch_size)
                b = self.frames.index_select(0, self._b_local[sl].reshape(-1))
                a = self.frames.index_select(0, self._a_local[sl].reshape(-1))
                if self.nf > 1:
                    b = b.view(batch_size, self.nf, 224, 224, 3)
                    a = a.view(batch_size, self.nf, 224, 224, 3)
                i = self._idx_dev[sl]
                yield (b, a, 0, self.act.index_select(0, i), self.rew.index_select(0, i), self.term.index_select(0, i),
                       self.valid.index_select(0, i), self.gt.index_select(0, i))
            epoch += 1
            print("reset iterator")


class HostFrameStream:
    """Streaming input path for decoded-frame shards that do NOT fit in HBM ("per-rank mmap + pinned
    double-buffer H2D"; replaces dataloaders/q_learning_real.py:55-73 under torch's DataLoader, train_q_network.py:98,114).

    No worker processes and no shared-memory hop: the shards stay memory-mapped in THIS process; a producer thread computes each
    minibatch's frame addresses (the PREVIOUS_IMAGES gather is index arithmetic on the shard index), has ``vdqn_host_gather``
    (include/vdqn.h; a few native threads, GIL released) copy the frames ONCE from the page cache into one of ``depth`` pinned
    staging buffers, and queues the host-to-device copy on a prefetch stream; the training loop finds the batch on the device
    with an event to wait for.  Minibatch order and content are those of ``DeviceFrameStore.batches`` for the same seed (one
    seeded permutation per epoch, rank r takes perm[r::world], drop_last), bit for bit — tests/test_shards_cpu.py."""

    def __init__(self, location, device, batch_size: int, seed: int, rank: int = 0, world_size: int = 1, threads: int = 0, depth: int = 3,
                 **flags):
        import ctypes
        import queue
        import threading
        from . import _lib
        ds = ShardDataset(location, **flags)
        self.flags = ds
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda and self.device.index is None:  # the producer thread needs an explicit device
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.B, self.seed, self.rank, self.world = int(batch_size), int(seed), int(rank), int(world_size)
        self.nf = 4 if ds.previous_images else 1
        self.n = len(ds)
        self.per_rank = (self.n // self.world // self.B) * self.B
        if self.per_rank == 0:
            raise ValueError(f"dataset of {self.n} samples is smaller than one global batch ({self.B} x {self.world})")
        self._maps = [np.load(p, mmap_mode="r") for p in ds._paths]
        self._base = np.array([m.ctypes.data for m in self._maps], dtype=np.uint64)  # address of frame 0 of every shard
        self._shard_frames = ds.shard_frames
        self._before = np.ascontiguousarray(ds.before[:, :self.nf].astype(np.int64))
        self._after = np.ascontiguousarray(ds.after[:, :self.nf].astype(np.int64))
        lab = DeviceFrameStore._labels(ds, np.arange(self.n))
        self.act, self.rew, self.term, self.gt, self.valid = (x.to(self.device) for x in lab)
        self.threads = int(threads) if threads else max(1, min(16, (os.cpu_count() or 2) // 2))
        self._lib, self._C = _lib.load(), ctypes
        frames = 2 * self.B * self.nf  # before + after
        self._slots = []
        for _ in range(max(2, int(depth))):
            t = torch.empty((frames, 224, 224, 3), dtype=torch.uint8)
            self._slots.append(t.pin_memory() if self.cuda else t)
        self._slot_free = [None] * len(self._slots)  # event behind the last host-to-device copy that read the slot
        self._stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._q = queue.Queue(maxsize=len(self._slots) - 1)
        self._stop = False
        self._err = None
        self._thread = threading.Thread(target=self._produce, name="vdqn-host-frame-stream", daemon=True)
        self._thread.start()

    def __len__(self):
        return self.n

    def _perm(self, epoch: int) -> np.ndarray:
        g = torch.Generator(device="cpu")
        g.manual_seed(self.seed + epoch)
        return torch.randperm(self.n, generator=g)[self.rank::self.world][:self.per_rank].numpy()

    def _addresses(self, idx: np.ndarray) -> np.ndarray:
        fr = np.concatenate([self._before[idx].reshape(-1), self._after[idx].reshape(-1)])
        return self._base[fr // self._shard_frames] + (fr % self._shard_frames).astype(np.uint64) * np.uint64(FRAME_BYTES)

    def _produce(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            epoch, k = 0, 0
            while not self._stop:
                perm = self._perm(epoch)
                for lo in range(0, self.per_rank, self.B):
                    if self._stop:
                        return
                    idx = perm[lo:lo + self.B]
                    slot = k % len(self._slots)
                    k += 1
                    if self._slot_free[slot] is not None:
                        self._slot_free[slot].synchronize()  # the copy that read this staging buffer last has finished
                    buf = self._slots[slot]
                    addr = np.ascontiguousarray(self._addresses(idx))
                    rc = self._lib.vdqn_host_gather(buf.data_ptr(), addr.ctypes.data, addr.shape[0], FRAME_BYTES, self.threads)
                    if rc != 0:
                        raise RuntimeError("vdqn_host_gather failed")
                    idx_t = torch.from_numpy(idx)
                    if self.cuda:
                        with torch.cuda.stream(self._stream):
                            dev = buf.to(self.device, non_blocking=True)
                            idx_d = idx_t.to(self.device, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(self._stream)
                        self._slot_free[slot] = ev
                    else:
                        dev, idx_d, ev = buf.clone(), idx_t, None
                    if not self._put((dev, idx_d, ev)):
                        return
                epoch += 1
        except BaseException as e:  # noqa: BLE001 - handed to the consumer, which re-raises it
            self._err = e
            self._put(None)

    def _put(self, item) -> bool:
        """Queue ``item`` for the consumer; gives up (False) once close() has been called, so a full queue never holds the
        producer thread behind a consumer that has gone."""
        import queue
        while not self._stop:
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def batches(self):
        """Endless stream of device batches, the tuple of DeviceFrameStore.gather: (before, after, 0, act, rew, term, valid, gt)."""
        half = self.B * self.nf
        while True:
            item = self._q.get()
            if item is None:
                raise RuntimeError("the host frame stream stopped") from self._err
            dev, idx, ev = item
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                dev.record_stream(cur)  # allocated on the prefetch stream, consumed on the compute stream
                idx.record_stream(cur)
            b, a = dev[:half], dev[half:]
            if self.nf > 1:
                b = b.view(self.B, self.nf, 224, 224, 3)
                a = a.view(self.B, self.nf, 224, 224, 3)
            yield (b, a, 0, self.act.index_select(0, idx), self.rew.index_select(0, idx), self.term.index_select(0, idx),
                   self.valid.index_select(0, idx), self.gt.index_select(0, idx))

    def close(self, timeout: float = 30.0):
        """Stop the producer thread and JOIN it (it may be inside vdqn_host_gather, reading the memory maps, or queueing copies
        on the prefetch stream: neither may outlive this object), wait for the copies it queued, release the pinned slots."""
        import queue
        import time
        self._stop = True
        t = getattr(self, "_thread", None)
        deadline = time.monotonic() + timeout
        while t is not None and t.is_alive():
            try:
                while True:
                    self._q.get_nowait()
            except queue.Empty:
                pass
            t.join(0.05)
            if time.monotonic() > deadline:
                raise RuntimeError("the host frame stream's producer thread did not stop")
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        if self._stream is not None:
            self._stream.synchronize()
        self._slots, self._slot_free, self._maps = [], [], []
        self._thread = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            if getattr(self, "_thread", None) is not None:
                self.close(timeout=5.0)
        except Exception:  # noqa: BLE001 - interpreter teardown
            pass


def collate_batches(items):
    """collate_fn for ShardDataset.__getitems__: the fetch already produced the collated batch."""
    return items[0]


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit("usage: python -m video_dqn_amd.shards <data.feather> <out_dir> [frames_per_shard]")
    info = build_shards(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2048)
    print(info)
