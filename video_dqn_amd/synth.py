"""Deterministic, torch-RNG-independent synthetic weights / frames / minibatches.

Both sides of every parity test (the CPU oracle here, the HIP path on the GPU box) and the golden
generator regenerate identical tensors from ``(seed, name)`` with a counter-based splitmix64
hash, so no weight file has to be committed or shipped.  The shapes follow the reference:
``archs/HabitatDQNMultiAction.py:9-34`` (model), ``dataloaders/q_learning_real.py:55-98`` (batch
tuple), ``util/torch.py:5-12`` (ImageNet normalisation).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def bits(seed: int, name: str, n: int) -> np.ndarray:
    """n pseudo-random uint64 words for stream (seed, name)."""
    base = np.uint64((_fnv1a(name) ^ (seed * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + base
    return _splitmix(_splitmix(ctr))


def uniform(seed: int, name: str, shape, lo=0.0, hi=1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(seed, name, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def randint(seed: int, name: str, shape, n_values: int) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return ((bits(seed, name, n) >> np.uint64(33)) % np.uint64(n_values)).astype(np.int64).reshape(shape)


# ---------------------------------------------------------------------------------------------
# model weights
# ---------------------------------------------------------------------------------------------
def resnet18_param_shapes():
    """(name, shape, kind) in torchvision resnet18 ``state_dict()`` order (kinds: conv, bn_w, bn_b,
    bn_mean, bn_var, bn_nbt, fc_w, fc_b)."""
    out = []

    def bn(prefix, c):
        out.extend([(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
                    (prefix + ".running_mean", (c,), "bn_mean"), (prefix + ".running_var", (c,), "bn_var"),
                    (prefix + ".num_batches_tracked", (), "bn_nbt")])

    out.append(("conv1.weight", (64, 3, 7, 7), "conv"))
    bn("bn1", 64)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f"layer{li}.{bi}"
            stride = 2 if (li > 1 and bi == 0) else 1
            out.append((p + ".conv1.weight", (planes, inpl, 3, 3), "conv"))
            bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv"))
            bn(p + ".bn2", planes)
            if stride != 1 or inpl != planes:
                out.append((p + ".downsample.0.weight", (planes, inpl, 1, 1), "conv"))
                bn(p + ".downsample.1", planes)
            inpl = planes
    out.append(("fc.weight", (1000, 512), "fc_w"))
    out.append(("fc.bias", (1000,), "fc_b"))
    return out


# index of each resnet child inside ``features`` (children()[:-2] / [:-1])
_FEATURE_INDEX = {"conv1": 0, "bn1": 1, "layer1": 4, "layer2": 5, "layer3": 6, "layer4": 7}


def make_state_dict(seed: int, action_dim=3, num_classes=5, extra_capacity=True, num_frames=1,
                    q_scale=0.08) -> "OrderedDict[str, torch.Tensor]":
    """A full ``HabitatDQNMultiAction.state_dict()`` (250 keys for extra_capacity, 244 for basic;
    ``features.N.*`` alias the same tensors as ``resnet.*``) with non-trivial BN statistics."""
    res = OrderedDict()
    for name, shape, kind in resnet18_param_shapes():
        if kind == "conv":
            fan_in = shape[1] * shape[2] * shape[3]
            a = (6.0 / fan_in) ** 0.5  # uniform with var 2/fan_in
            t = uniform(seed, name, shape, -a, a)
        elif kind == "bn_w":
            # keep the residual branch (bn2) small so 8 blocks do not blow the range up
            t = uniform(seed, name, shape, 0.25, 0.55) if name.endswith("bn2.weight") else \
                uniform(seed, name, shape, 0.6, 1.2)
        elif kind == "bn_b":
            t = uniform(seed, name, shape, -0.2, 0.3)
        elif kind == "bn_mean":
            t = uniform(seed, name, shape, -0.3, 0.3)
        elif kind == "bn_var":
            t = uniform(seed, name, shape, 0.5, 1.5)
        elif kind == "bn_nbt":
            t = np.array(0, dtype=np.int64)
        elif kind == "fc_w":
            t = uniform(seed, name, shape, -0.04, 0.04)
        else:
            t = uniform(seed, name, shape, -0.04, 0.04)
        res[name] = torch.tensor(0, dtype=torch.int64) if kind == "bn_nbt" else torch.from_numpy(np.ascontiguousarray(t))

    sd = OrderedDict()
    for k, v in res.items():
        sd["resnet." + k] = v
    for k, v in res.items():
        head = k.split(".")[0]
        if head in _FEATURE_INDEX:
            sd["features." + str(_FEATURE_INDEX[head]) + k[len(head):]] = v

    def lin(name, out_f, in_f, scale=1.0):
        a = scale * (3.0 / in_f) ** 0.5
        sd[name + ".weight"] = torch.from_numpy(uniform(seed, name + ".weight", (out_f, in_f), -a, a))
        sd[name + ".bias"] = torch.from_numpy(uniform(seed, name + ".bias", (out_f,), -0.05, 0.15))

    if extra_capacity:
        a = (6.0 / (512 * 9)) ** 0.5
        sd["features.8.weight"] = torch.from_numpy(uniform(seed, "features.8.weight", (64, 512, 3, 3), -a, a))
        sd["features.8.bias"] = torch.from_numpy(uniform(seed, "features.8.bias", (64,), -0.1, 0.1))
        lin("top.0", 512, 1600 * num_frames, 1.4)
        lin("top.2", 256, 512, 1.4)
        lin("top.4", action_dim * num_classes, 256, q_scale)
    else:
        lin("top", action_dim * num_classes, 512 * num_frames, q_scale)
    return sd


# ---------------------------------------------------------------------------------------------
# frames and minibatches
# ---------------------------------------------------------------------------------------------
def make_frames_uint8(seed: int, name: str, batch: int, num_frames: int = 1, size: int = 224,
                      structured: bool = False) -> np.ndarray:
    """uint8 NHWC frames [B, F, size, size, 3], i.i.d. U{0..255}.

    ``structured=True`` (parity tests) mixes the noise with a per-frame colour offset and two
    smooth ramps so different samples produce visibly different Q-values."""
    n = batch * num_frames * size * size * 3
    words = bits(seed, name, (n + 7) // 8)
    noise = words.view(np.uint8)[:n].reshape(batch, num_frames, size, size, 3).copy()
    if not structured:
        return noise
    k = batch * num_frames
    off = uniform(seed, name + ".off", (k, 1, 1, 3), 0.0, 160.0)
    gx = uniform(seed, name + ".gx", (k, 1, 1, 3), -90.0, 90.0)
    gy = uniform(seed, name + ".gy", (k, 1, 1, 3), -90.0, 90.0)
    amp = uniform(seed, name + ".amp", (k, 1, 1, 1), 0.1, 0.6)
    r = np.linspace(0.0, 1.0, size, dtype=np.float32)
    img = (noise.reshape(k, size, size, 3).astype(np.float32) * amp + off
           + gx * r[None, None, :, None] + gy * r[None, :, None, None])
    return np.clip(np.rint(img), 0, 255).astype(np.uint8).reshape(batch, num_frames, size, size, 3)


def normalise_frames(frames_u8: np.ndarray) -> torch.Tensor:
    """uint8 [B,F,H,W,3] -> float32 [B,F,3,H,W] (or [B,3,H,W] when F == 1), exactly as
    ``ToTensor`` + ``Normalize`` of util/torch.py:5-12 do: (x/255 - mean)/std in fp32."""
    x = torch.from_numpy(frames_u8).float().div(255.0)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32)
    x = ((x - mean) / std).permute(0, 1, 4, 2, 3).contiguous()
    return x[:, 0] if x.shape[1] == 1 else x


def make_batch(seed: int, batch: int, num_frames: int = 1, action_dim: int = 3, reward_p: float = 0.05,
               size: int = 224, structured: bool = False):
    """The 7-tuple a collated ``QLearningRealDataset`` batch holds (q_learning_real.py:98):
    before, after (fp32 normalised), act i64 [B], rew i64 [B,5], term (= rew), gt f64 [B] NaN,
    valid_mask i64 [B,5].  Also returns the raw uint8 frames for the fused-normalise input path."""
    fb = make_frames_uint8(seed, "before", batch, num_frames, size, structured)
    fa = make_frames_uint8(seed, "after", batch, num_frames, size, structured)
    act = torch.from_numpy(randint(seed, "act", (batch,), action_dim))
    rew = torch.from_numpy((uniform(seed, "rew", (batch, 5)) < reward_p).astype(np.int64))
    term = rew.clone()
    gt = torch.full((batch,), float("nan"), dtype=torch.float64)
    valid = torch.ones((batch, 5), dtype=torch.int64)
    tup = (normalise_frames(fb), normalise_frames(fa), act, rew, term, gt, valid)
    return tup, (fb, fa)


def make_inverse_state_dict(seed: int) -> "OrderedDict[str, torch.Tensor]":
    """A full state_dict of the inverse-action model (archs/inverse_action2.py:45-70: ``resnet18.<i>.*`` for the
    Sequential over the ResNet children [:-2], then conv1..3, fc1, fc2, fc_accuracy), deterministic like make_state_dict."""
    base = make_state_dict(seed)
    seq = {"conv1": 0, "bn1": 1, "layer1": 4, "layer2": 5, "layer3": 6, "layer4": 7}
    sd = OrderedDict()
    for k, v in base.items():
        if not k.startswith("resnet.") or k.startswith("resnet.fc."):
            continue
        parts = k.split(".")
        sd["resnet18." + ".".join([str(seq[parts[1]])] + parts[2:])] = v

    def conv(name, co, ci, ksz):
        a = (6.0 / (ci * ksz * ksz)) ** 0.5
        sd[name + ".weight"] = torch.from_numpy(uniform(seed, "inv." + name + ".weight", (co, ci, ksz, ksz), -a, a))
        sd[name + ".bias"] = torch.from_numpy(uniform(seed, "inv." + name + ".bias", (co,), -0.1, 0.1))

    def lin(name, out_f, in_f, scale=1.0):
        a = scale * (3.0 / in_f) ** 0.5
        sd[name + ".weight"] = torch.from_numpy(uniform(seed, "inv." + name + ".weight", (out_f, in_f), -a, a))
        sd[name + ".bias"] = torch.from_numpy(uniform(seed, "inv." + name + ".bias", (out_f,), -0.1, 0.1))
    conv("conv1", 256, 1024, 1)
    conv("conv2", 256, 256, 3)
    conv("conv3", 64, 256, 3)
    lin("fc1", 128, 576, 1.4)
    lin("fc2", 3, 128, 1.4)
    lin("fc_accuracy", 3, 3, 1.0)
    return sd
