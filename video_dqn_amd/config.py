"""Experiment configuration with the reference's surface:

  * ``get_cfg_defaults()`` — the keys and defaults of the reference's ``defaults.py:7-37`` (a yacs ``CfgNode``
    there; yacs is not installed here, so ``CfgNode`` below re-implements the subset the reference uses:
    attribute/item access, ``clone``, ``merge_from_file`` that rejects unknown keys and mismatched types like
    yacs does, ``freeze``).
  * ``ExperimentConfig(folder, device=None, remove=False, resume=False, run_prefix='run', tensorboard=True)``
    — ``experiment_config.py:12-51``: run-directory numbering, summary writer, ``<folder>/config.yml`` merged
    over the defaults, LOSS_CLIP validation, every key copied onto the instance, ``device``/``gpu_id``.

Keys added by this build (all optional, defaults reproduce the reference): BATCH_SIZE (the reference
hard-codes 16, train_q_network.py:98), NUM_WORKERS (8), COMPUTE_DTYPE ('bf16' | 'f32'), NUM_FRAMES (0 = the
reference's rule: 4 if PANORAMA or PREVIOUS_IMAGES else 1), SYNTHETIC_DATA (train on generated frames), DEVICE_RESIDENT_DATA ('auto' | 'on' | 'off': keep a decoded-frame shard
dataset in HBM and gather minibatches on the device), SHARD_INPUT ('stream' | 'dataloader': how a shard dataset that is NOT
resident reaches the GPU — memory-mapped shards + native gather into pinned double buffers, or torch's DataLoader),
HOST_GATHER_THREADS, SYNC_BN (ARCHITECTURE='basic' on several GPUs: global BatchNorm
statistics, so N ranks equal the reference's single big batch; default True), DETERMINISTIC (run-to-run bit-identical
updates — the reference's cudnn.deterministic = True, train_q_network.py:88-89 — at a small throughput cost), LOSS_KIND ('l2' = the reference's half
squared TD error, train_q_network.py:167; 'huber' = smooth-L1, the option archs/HabitatDQNMultiAction.py:25 leaves open),
PRETRAINED_WEIGHTS (path of a torchvision resnet18 state_dict: the reference builds models.resnet18(pretrained=True),
archs/HabitatDQNMultiAction.py:11, which needs a download this build cannot make), BOOTSTRAP_CHECKPOINT (the file the
BOOTSTRAP branch loads; default = the path hard-coded at train_q_network.py:202).
"""
from __future__ import annotations

import copy
import json
import os
import re
import shutil

import yaml

VALID_VALUES = {"LOSS_CLIP": ["sigmoid", "rect", "none"],  # experiment_config.py:10
                "LOSS_KIND": ["l2", "huber"]}


class CfgNode(dict):
    """Minimal yacs-compatible node (flat: the reference's config has no nesting)."""

    def __init__(self, init=None):
        super().__init__(init or {})
        object.__setattr__(self, "_frozen", False)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self._frozen:
            raise AttributeError(f"Attempted to set {k} to {v}, but CfgNode is immutable")
        self[k] = v

    def clone(self):
        c = CfgNode(copy.deepcopy(dict(self)))
        return c

    def freeze(self):
        object.__setattr__(self, "_frozen", True)

    def defrost(self):
        object.__setattr__(self, "_frozen", False)

    def merge_from_file(self, path):
        with open(path) as f:
            loaded = yaml.safe_load(f) or {}
        self.merge_from_dict(loaded)

    def merge_from_dict(self, loaded):
        for k, v in loaded.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {k}")
            self[k] = _check_and_coerce(v, self[k], k)

    def dump(self):
        return yaml.safe_dump(dict(self))

    def __str__(self):
        return "\n".join(f"{k}: {self[k]!r}" if isinstance(self[k], str) else f"{k}: {self[k]}" for k in sorted(self))


def _check_and_coerce(new, old, key):
    """yacs' _check_and_coerce_cfg_value_type: same type, or int<->float / tuple<->list casts, else error."""
    if type(new) is type(old):
        return new
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, int) and not isinstance(old, bool) and isinstance(new, float) and float(new).is_integer():
        return int(new)
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new)
    raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) with values ({old} vs. {new}) for config key: {key}")


def get_cfg_defaults() -> CfgNode:
    """defaults.py:7-37 (+ this build's optional keys)."""
    c = CfgNode()
    c.PANORAMA = True
    c.SEED = 0
    c.TRAIN_ON_GROUND_TRUTH = False
    c.DATASET = "none"
    c.SUB_DATASET = "none"
    c.CLASS_LABEL = "toilet"
    c.LOSS_CLIP = "none"
    c.ARCHITECTURE = "basic"
    c.RANDOM_ACTIONS = False
    c.ONE_ACTION = False
    c.SEMANTIC_REWARDS = False
    c.DETECTION_REWARDS = False
    c.REMOVE_BEFORE_REWARD = False
    c.USE_INVERSE_ACTIONS = False
    c.VALUE_LEARNING = False
    c.PREVIOUS_IMAGES = False
    c.GAMMA = 0.9
    c.BOOTSTRAP = False
    c.LINEAR = False
    c.LEARNING_RATE = 1e-3
    c.NUM_STEPS = int(1e5)
    c.TARGET_UPDATE_INTERVAL = int(8e3)
    c.CHECKPOINT_INTERVAL = int(2e3)
    c.ACTION_HIDDEN_LAYERS = 1
    c.GUMBEL_TEMP = 0.1
    c.CONFIDENCE_REWARD = False
    c.DISTRIBUTIONAL = False
    c.KL_BACKWARDS = False
    c.LOG_SIGMA = False
    c.VISUALIZATION_DATA_ROOT = ""
    # --- added by this build ---
    c.BATCH_SIZE = 16
    c.NUM_WORKERS = 8
    c.COMPUTE_DTYPE = "bf16"
    c.NUM_FRAMES = 0
    c.SYNTHETIC_DATA = False
    c.SYNC_BN = True
    c.DEVICE_RESIDENT_DATA = "auto"
    c.RANK_SHARDED_DATA = True    # resident data on several GPUs: every rank holds the frames of ITS samples of the epoch, not a full copy
    c.SHARD_INPUT = "stream"      # decoded-frame shards that are not resident: 'stream' (HostFrameStream) | 'dataloader'
    c.HOST_GATHER_THREADS = 0     # native threads of the streaming path's frame gather (0 = min(16, cores / 2))
    c.DETERMINISTIC = False
    c.LOSS_KIND = "l2"
    c.PRETRAINED_WEIGHTS = ""
    c.BOOTSTRAP_CHECKPOINT = "logs/trained_gt_0.99/models/epoch99.torch"
    return c


class JsonlWriter:
    """Stand-in for tensorboardX.SummaryWriter when it is not installed: same add_scalar/add_image/close calls,
    scalars appended to <log_dir>/scalars.jsonl."""

    def __init__(self, log_dir, comment=""):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, global_step=None):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": global_step}) + "\n")
        self._f.flush()

    def add_image(self, *a, **k):
        pass

    def close(self):
        self._f.close()


def _make_writer(log_dir):
    try:
        from tensorboardX import SummaryWriter  # the reference's writer (experiment_config.py:5,32)
        return SummaryWriter(log_dir=log_dir, comment=log_dir)
    except ImportError:
        try:
            from torch.utils.tensorboard import SummaryWriter
            return SummaryWriter(log_dir=log_dir, comment=log_dir)
        except Exception:
            return JsonlWriter(log_dir)


class ExperimentConfig:
    """experiment_config.py:12-51."""

    def __init__(self, folder, device=None, remove=False, resume=False, run_prefix="run", tensorboard=True):
        import torch
        self.folder = folder
        if remove:  # :16-17  rm -r <folder>/<run_prefix>*
            for f in os.listdir(folder):
                if f.startswith(run_prefix):
                    p = os.path.join(folder, f)
                    shutil.rmtree(p) if os.path.isdir(p) else os.remove(p)
        self.files = sorted(os.listdir(folder))  # :19  (`ls` order)
        max_run = 0
        for f in self.files:  # :21-24  last match wins
            m = re.search(f"^{run_prefix}(\\d+)$", f)
            if m:
                max_run = int(m[1])
        if not resume:
            max_run += 1
        log_dir = f"{folder}/{run_prefix}{max_run}"
        self.log_dir = log_dir
        if tensorboard:
            self.writer = _make_writer(log_dir)
        self.cfg = get_cfg_defaults()
        self.cfg.merge_from_file(f"{folder}/config.yml")
        self.cfg.freeze()
        for k in VALID_VALUES:  # :37-39
            if self.cfg[k] not in VALID_VALUES[k]:
                raise Exception(f"Invalid value for {k}")
        for k in self.cfg:  # :41-42
            setattr(self, k, self.cfg[k])
        self.gpu_id = None
        if device is not None:  # :45-51
            self.device = torch.device(device)
            m = re.match(r".*:(\d+)", str(device))
            if m:
                self.gpu_id = int(m[1])
        else:
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
