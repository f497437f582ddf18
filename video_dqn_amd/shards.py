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


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit("usage: python -m video_dqn_amd.shards <data.feather> <out_dir> [frames_per_shard]")
    info = build_shards(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2048)
    print(info)
