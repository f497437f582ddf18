#!/usr/bin/env python
"""Input-side throughput of the decoded-frame shard path: samples/s a DataLoader over ShardDataset
delivers as pinned uint8 batches, against what the GPU path consumes (~34 k tuples/s = 68 k frames/s per GPU).
Builds a synthetic shard directory (random frames, no JPEG decoding involved) under /tmp and times the loader alone."""
import os
import sys
import time

import numpy as np
import torch
from torch.utils import data

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_dqn_amd.shards import ShardDataset, collate_batches  # noqa: E402


def make_shards(root, n_frames=4096, shard_frames=1024, n_samples=16384):
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(0)
    for s in range(n_frames // shard_frames):
        mm = np.lib.format.open_memmap(os.path.join(root, f"frames_{s:05d}.npy"), mode="w+", dtype=np.uint8, shape=(shard_frames, 224, 224, 3))
        mm[:] = rng.integers(0, 256, (shard_frames, 224, 224, 3), dtype=np.uint8)
        mm.flush()
    before = rng.integers(0, n_frames, (n_samples, 4))
    after = rng.integers(0, n_frames, (n_samples, 4))
    np.savez(os.path.join(root, "index.npz"), before=before, after=after, shard_frames=np.int64(shard_frames), n_frames=np.int64(n_frames),
             detector_score=rng.random((n_samples, 5)), sparse_reward=rng.integers(0, 2, (n_samples, 5)),
             steps_to_reward=rng.random((n_samples, 5)), inverse_actions=rng.integers(0, 3, n_samples),
             has_inverse_actions=np.int64(1), with_previous=np.int64(1))


def run(root, workers, batch=256, batches=40, batched_fetch=True):
    ds = ShardDataset(root, inverse_actions=True)
    kw = dict(batch_size=batch, shuffle=True, drop_last=True, num_workers=workers, pin_memory=True, persistent_workers=workers > 0)
    if batched_fetch:
        ds.batched_fetch = True
        kw["collate_fn"] = collate_batches
    loader = data.DataLoader(ds, **kw)
    it = iter(loader)
    for _ in range(3):
        next(it)
    t0 = time.perf_counter()
    n = 0
    for _ in range(batches):
        b = next(it)
        n += b[0].shape[0]
    dt = time.perf_counter() - t0
    return n / dt


if __name__ == "__main__":
    root = "/tmp/vdqn_loader_shards"
    if not os.path.exists(os.path.join(root, "index.npz")):
        make_shards(root)
    for w in (4, 8, 16, 32):
        print(f"workers {w:2d}: {run(root, w):9.0f} samples/s (batched fetch)   {run(root, w, batched_fetch=False):9.0f} samples/s (per-sample fetch)", flush=True)
