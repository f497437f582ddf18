"""The (s, s', a, r, term, gt, valid_mask) stream of the reference's Q-learning loader.

``QLearningRealDataset`` mirrors ``dataloaders/q_learning_real.py:27-98`` on the same feather schema
(``before_image, after_image, im_start, detector_score0..4, sparse_reward0..4, steps_to_reward0..4,
inverse_actions`` — written by ``dataset/process_episodes_real.py:144-181``).  Differences, all opt-in:

  * ``as_uint8=True`` returns the resized + centre-cropped frame as uint8 HWC (150 KB instead of 602 KB per
    frame); the (x/255 - mean)/std of ``util/torch.py:5-12`` is then fused into the GPU input-packing kernel.
  * the reference's ``np.int`` (:82) no longer exists in numpy >= 1.24; ``np.int64`` is what it aliased.

``SyntheticTupleDataset`` generates the SURVEY §8(d) synthetic stream (U{0..255} frames, act ~ U{0,1,2},
rew ~ Bernoulli(0.05) per category, term = rew) for benchmarks and for running the trainer without data.
"""
from __future__ import annotations

import re

import numpy as np
import torch
from torch.utils import data

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

# Confidence thresholds (dataloaders/q_learning_real.py:15-18)
detection_thresholds = [0.9700177907943726, 0.9738382697105408, 0.9512060284614563,
                        0.7334915995597839, 0.7058018445968628]


def multi_get(df, name):
    """util/pd.py:10-14: the columns ``<name>0..<name>k`` of a frame or a row as one array."""
    import pandas as pd
    cols = df.keys() if df.__class__ == pd.Series else df.columns
    num_cols = len([c for c in cols if re.match(f"{name}\\d+", c)])
    sc = df[[f"{name}{c}" for c in range(num_cols)]]
    return np.array(tuple(sc) if df.__class__ == pd.Series else sc)


def resize_center_crop_u8(img, size=224):
    """transforms.Resize(size) + CenterCrop(size) of util/torch.py:5-12 on a PIL image -> uint8 HWC array.
    (Resize: shorter side to ``size`` with bilinear interpolation, long side int(size * long / short).)"""
    from PIL import Image
    img = img.convert("RGB")
    w, h = img.size
    if (w <= h and w != size) or (h <= w and h != size):
        if w < h:
            ow, oh = size, int(size * h / w)
        else:
            oh, ow = size, int(size * w / h)
        img = img.resize((ow, oh), Image.BILINEAR)
    w, h = img.size
    left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    return np.asarray(img, dtype=np.uint8)


def image_net_transform(img_u8):
    """ToTensor + Normalize(ImageNet) (util/torch.py:9-11): uint8 HWC -> float32 CHW."""
    x = torch.from_numpy(img_u8).float().div(255.0)
    x = (x - torch.tensor(IMAGENET_MEAN)) / torch.tensor(IMAGENET_STD)
    return x.permute(2, 0, 1).contiguous()


def tuple_columns(samples, gamma, value_learning=False, confidence_reward=False, inverse_actions=False, one_action=False,
                  slam_actions=False):
    """The non-image part of every tuple of a feather table, computed once for the whole table (the reference derives it row by
    row inside __getitem__, dataloaders/q_learning_real.py:72-98; shards.write_shards stores the same columns): returns
    (reward [N,5], gt [N,5] or None, action [N]).

      reward : detector_score > per-category threshold as int64 (:82), or the raw scores under CONFIDENCE_REWARD (:78-80)
      gt     : gamma ** steps_to_reward with inf -> NaN under VALUE_LEARNING (:85-87), else the scalar NaN
      action : the `inverse_actions` column (:89-90) or zeros under ONE_ACTION (:93-94); SLAM actions were never implemented"""
    det = multi_get(samples, "detector_score")
    reward = det if confidence_reward else (det > detection_thresholds).astype(np.int64)
    gt = None
    if value_learning:
        steps = multi_get(samples, "steps_to_reward")
        gt = np.power(np.full(steps.shape, gamma), steps)
        gt[steps == np.inf] = np.nan
    if inverse_actions:
        action = samples["inverse_actions"].to_numpy()
    elif slam_actions:
        raise NotImplementedError("not implemented")  # :91-92
    elif one_action:
        action = np.zeros(len(samples), dtype=np.int64)
    else:
        raise Exception("not implemented")  # :95-96
    return reward, gt, action


class QLearningRealDataset(data.Dataset):
    """Same constructor arguments and the same 7-tuple per index as dataloaders/q_learning_real.py:27-98.  The table columns are
    turned into arrays once (``tuple_columns``); ``__getitem__`` only decodes the two frames (or 2 x 4 under PREVIOUS_IMAGES)
    and indexes those arrays."""

    _FRAME = re.compile(r"(.*?/)(\d+).jpg")

    def __init__(self, location=None, one_action=False, value_learning=False, inverse_actions=False,
                 previous_images=False, confidence_reward=False, slam_actions=False, gamma=0.99, as_uint8=False):
        import pandas as pd
        self.samples = pd.read_feather(location)  # :37
        self.gamma, self.as_uint8, self.previous_images = gamma, as_uint8, previous_images
        self.value_learning, self.confidence_reward = value_learning, confidence_reward
        self.slam_actions, self.one_action, self.inverse_actions = slam_actions, one_action, inverse_actions
        self._mode_error = None
        try:
            self._reward, self._gt, self._action = tuple_columns(self.samples, gamma, value_learning, confidence_reward,
                                                                 inverse_actions, one_action, slam_actions)
        except Exception as e:  # the reference raises at the first __getitem__, not in the constructor
            if len(e.args) != 1 or e.args[0] != "not implemented":
                raise
            self._mode_error = e
        self._paths = (self.samples["before_image"].to_numpy(), self.samples["after_image"].to_numpy())
        self._start = self.samples["im_start"].to_numpy() if previous_images else None

    def __len__(self):
        return len(self.samples)

    def reward_percentage(self):  # :50-52
        rewards = multi_get(self.samples, "sparse_reward")
        return (rewards.max(axis=1) > 0).sum() / rewards.shape[0]

    def _load(self, path):
        from PIL import Image
        u8 = resize_center_crop_u8(Image.open(path))
        return torch.from_numpy(u8.copy()) if self.as_uint8 else image_net_transform(u8)

    def _frames(self, path, index):
        if not self.previous_images:
            return self._load(path)
        # :60-67  frames id, id-1, id-2, id-3 of the same episode folder, clamped at the episode's first frame
        m = self._FRAME.match(path)
        first = self._start[index]
        return torch.stack([self._load(m[1] + "%04d.jpg" % max(int(m[2]) - back, first)) for back in range(4)])

    def __getitem__(self, index):  # :55-98
        if self._mode_error is not None:
            raise self._mode_error
        before, after = (self._frames(col[index], index) for col in self._paths)
        reward = self._reward[index]
        gt = self._gt[index] if self._gt is not None else np.nan
        # (reward twice: the reference's `termainl` typo, :80, leaves terminal = reward in every mode; :98)
        return before, after, self._action[index], reward, reward, gt, np.ones_like(reward)


class SyntheticTupleDataset(data.Dataset):
    """Endless-ish synthetic tuples (deterministic per index)."""

    def __init__(self, length=4096, num_frames=1, action_dim=3, reward_p=0.05, seed=0, as_uint8=True):
        self.length, self.num_frames, self.action_dim, self.reward_p, self.seed = length, num_frames, action_dim, reward_p, seed
        self.as_uint8 = as_uint8

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        g = np.random.default_rng(self.seed * 1000003 + index)
        shape = (224, 224, 3) if self.num_frames == 1 else (self.num_frames, 224, 224, 3)
        fb = g.integers(0, 256, shape, dtype=np.uint8)
        fa = g.integers(0, 256, shape, dtype=np.uint8)
        if self.as_uint8:
            bi, ai = torch.from_numpy(fb), torch.from_numpy(fa)
        else:
            f = (lambda a: image_net_transform(a)) if self.num_frames == 1 else (lambda a: torch.stack([image_net_transform(x) for x in a]))
            bi, ai = f(fb), f(fa)
        reward = (g.random(5) < self.reward_p).astype(np.int64)
        return bi, ai, int(g.integers(0, self.action_dim)), reward, reward, np.nan, np.ones_like(reward)
