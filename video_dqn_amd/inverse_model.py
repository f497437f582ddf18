"""The inverse-action model of the reference (``archs/inverse_action2.py:45-100``) for INFERENCE on the HIP kernels —
the use the Q-learning data pipeline makes of it: ``dataset/process_episodes_real.py:84-95,164-179`` loads
``inverse_model.torch``, puts the model in eval mode and labels every (before, after) frame pair with
``model(be, ae)[1].argmax(1)``; those labels are the ``inverse_actions`` column the Q-learning loader reads
(``dataloaders/q_learning_real.py``).

    m = InverseActionModel(dtype="bf16", device="cuda"); m.load_state_dict(torch.load("inverse_model.torch")); m.eval()
    encoding, y = m(be, ae)                 # float[B,3,224,224] normalised tensors, or uint8 [B,224,224,3] frames
    actions = y.argmax(dim=1)

* state_dict keys are the reference's (``resnet18.<i>.*`` = the Sequential over the ResNet children, ``conv1..3``,
  ``fc1``, ``fc2``, ``fc_accuracy``), so the published checkpoint loads unchanged (strict).
* The frozen trunk runs through ``vdqn_net_trunk_forward`` (both frames in one 2B pass), the head through ``vdqn_conv2d``
  (1x1 over the 1024-channel concat, two valid 3x3 convs, three linears) and ``vdqn_softmax_rows``.  torch is used for
  device memory and for re-laying-out the head weights once per load; there is no CPU arithmetic and no fallback.
* Training of this model: ``video_dqn_amd/inverse_train.py`` (``train_inverse_model.py``'s loop on the same kernels).
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib, ops
from .engine import NetEngine, _ptr, _stream

# resnet18.<i> (nn.Sequential over list(resnet.children())[:-2], archs/inverse_action2.py:50-52) -> engine trunk names
_SEQ = {0: "resnet.conv1", 1: "resnet.bn1", 4: "resnet.layer1", 5: "resnet.layer2", 6: "resnet.layer3", 7: "resnet.layer4"}


def _trunk_name(key: str):
    parts = key.split(".")
    idx = int(parts[1])
    if idx not in _SEQ:
        return None
    return ".".join([_SEQ[idx]] + parts[2:])


class InverseActionModel(nn.Module):
    HEAD = (("conv1", (256, 1024, 1, 1)), ("conv2", (256, 256, 3, 3)), ("conv3", (64, 256, 3, 3)),
            ("fc1", (128, 576)), ("fc2", (3, 128)), ("fc_accuracy", (3, 3)))

    def __init__(self, dtype=None, device=None, max_batch=32):
        super().__init__()
        dtype = dtype or os.environ.get("VDQN_DTYPE", "bf16")
        # the trunk lives in a HabitatDQN engine instance (its Q-head stays unused)
        self.engine = NetEngine(3, 5, 1, True, dtype, 2 * max_batch, device)
        self.max_batch = max_batch
        dev = self.engine.device
        self.tdtype = torch.bfloat16 if self.engine.dtype_name == "bf16" else torch.float32
        self._trunk_keys = OrderedDict()  # reference key -> engine slot name
        for name in self.engine.slots:
            if not name.startswith("resnet.") or name.startswith("resnet.fc."):
                continue
            parts = name.split(".")
            base = {v: k for k, v in _SEQ.items()}[".".join(parts[:2])]
            self._trunk_keys["resnet18." + ".".join([str(base)] + parts[2:])] = name
        self.head = OrderedDict()
        for n, shape in self.HEAD:
            self.head[n + ".weight"] = torch.zeros(shape, dtype=torch.float32, device=dev)
            self.head[n + ".bias"] = torch.zeros(shape[0], dtype=torch.float32, device=dev)
        self._packed_head = None

    # ---- reference state_dict layout ---------------------------------------------------------------------
    def state_dict(self, *a, **k):
        sd = OrderedDict()
        order = []
        for i in (0, 1, 4, 5, 6, 7):
            order += [k_ for k_ in self._trunk_keys if k_.split(".")[1] == str(i)]
        tv_order = _torchvision_order(order)
        for key in tv_order:
            sd[key] = self.engine.view(self._trunk_keys[key]).detach()
            if key.endswith("running_var"):
                sd[key[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long, device=self.engine.device)
        for key, t in self.head.items():
            sd[key] = t
        return sd

    def load_state_dict(self, state_dict, strict=True):
        want = set(self.state_dict().keys())
        got = set(state_dict.keys())
        if strict and want != got:
            raise RuntimeError(f"InverseActionModel.load_state_dict: missing {sorted(want - got)[:4]} unexpected {sorted(got - want)[:4]}")
        with torch.no_grad():
            for key, slot in self._trunk_keys.items():
                if key in state_dict:
                    self.engine.view(slot).copy_(state_dict[key].to(torch.float32))
            for key in self.head:
                if key in state_dict:
                    self.head[key].copy_(state_dict[key].to(torch.float32))
        self.engine.mark_dirty()
        self._packed_head = None

    # ---- head weights in the kernels' layout ([co_pad][r][s][ci], K-contiguous) -----------------------------
    def _pack_head(self):
        dev, dt = self.engine.device, self.tdtype

        def conv_w(w):  # [co, ci, r, s] -> [co_pad, r, s, ci]
            co = w.shape[0]
            co_pad = (co + 63) // 64 * 64
            out = torch.zeros((co_pad, w.shape[2], w.shape[3], w.shape[1]), dtype=dt, device=dev)
            out[:co] = w.permute(0, 2, 3, 1).to(dt)
            return out.contiguous()

        def bias(b):
            out = torch.zeros(((b.numel() + 63) // 64 * 64,), dtype=torch.float32, device=dev)
            out[:b.numel()] = b
            return out
        h = self.head
        p = {"conv1": (conv_w(h["conv1.weight"]), bias(h["conv1.bias"])), "conv2": (conv_w(h["conv2.weight"]), bias(h["conv2.bias"])),
             "conv3": (conv_w(h["conv3.weight"]), bias(h["conv3.bias"]))}
        # fc1 consumes x.view(N, -1) of an NCHW [N,64,3,3] tensor (c*9 + hw); the kernels hold it as NHWC (hw*64 + c)
        w1 = h["fc1.weight"].view(128, 64, 9).permute(0, 2, 1).reshape(128, 576)
        p["fc1"] = (conv_w(w1.reshape(128, 576, 1, 1)), bias(h["fc1.bias"]))
        w2 = torch.zeros((64, 128), device=dev)
        w2[:3] = h["fc2.weight"]
        p["fc2"] = (conv_w(w2.reshape(64, 128, 1, 1)), bias(h["fc2.bias"]))
        w3 = torch.zeros((64, 64), device=dev)  # fc_accuracy reads the 3 fc2 outputs carried as 64 columns (zeros beyond 3)
        w3[:3, :3] = h["fc_accuracy.weight"]
        p["fc_accuracy"] = (conv_w(w3.reshape(64, 64, 1, 1)), bias(h["fc_accuracy.bias"]))
        self._packed_head = p

    # ---- forward (eval) -------------------------------------------------------------------------------------
    def forward(self, k, k_plus_one):  # archs/inverse_action2.py:72-100 (eval mode: dropout is the identity)
        eng = self.engine
        if self.training:
            raise _lib.VdqnError("InverseActionModel.forward is the eval-mode (labelling) path: call .eval(); training goes through inverse_train.InverseTrainer")
        if k.dtype == torch.uint8:
            src_kind, frames = 0, torch.cat([k, k_plus_one], 0)
        else:
            src_kind, frames = 1, torch.cat([k.float(), k_plus_one.float()], 0)
        B = k.shape[0]
        if B > self.max_batch:
            outs = [self.forward(k[i:i + self.max_batch], k_plus_one[i:i + self.max_batch]) for i in range(0, B, self.max_batch)]
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        frames = frames.to(eng.device).contiguous()
        with torch.cuda.device(eng.device):
            if eng._packed_version != eng.version_key():
                eng.pack_weights()
                eng._packed_version = eng.version_key()
            if self._packed_head is None:
                self._pack_head()
            acts = eng._acts_for(2 * B)
            _lib.check(eng.lib.vdqn_net_trunk_forward(eng.handle, _ptr(eng.packed), _ptr(frames), src_kind, 2 * B, _ptr(acts), _stream()),
                       "vdqn_net_trunk_forward")
            off = eng.lib.vdqn_net_act_offset(eng.handle, 2 * B, b"o7")
            esz = 2 if self.tdtype == torch.bfloat16 else 4
            feat = acts[off:off + 2 * B * 49 * 512 * esz].view(self.tdtype).view(2 * B, 7, 7, 512)
            x = torch.cat([feat[:B], feat[B:]], dim=3).contiguous()  # torch.cat([resnet_k, resnet_k_plus_one], dim=1) in NHWC
            p = self._packed_head
            x = ops.conv2d(x, p["conv1"][0], ho=7, wo=7, co=256, r=1, s=1, stride=1, pad=0, bias=p["conv1"][1], relu=True)
            x = ops.conv2d(x, p["conv2"][0], ho=5, wo=5, co=256, r=3, s=3, stride=1, pad=0, bias=p["conv2"][1], relu=True)
            x = ops.conv2d(x, p["conv3"][0], ho=3, wo=3, co=64, r=3, s=3, stride=1, pad=0, bias=p["conv3"][1], relu=True)
            x = x.view(B, 1, 1, 576)
            x = ops.conv2d(x, p["fc1"][0], ho=1, wo=1, co=128, r=1, s=1, stride=1, pad=0, bias=p["fc1"][1], relu=True)
            x, x32 = ops.conv2d(x, p["fc2"][0], ho=1, wo=1, co=64, r=1, s=1, stride=1, pad=0, bias=p["fc2"][1], want_f32=True)
            _, y32 = ops.conv2d(x, p["fc_accuracy"][0], ho=1, wo=1, co=64, r=1, s=1, stride=1, pad=0, bias=p["fc_accuracy"][1], want_f32=True)
            x32 = x32.view(B, 64)
            enc = torch.empty_like(x32)
            _lib.check(eng.lib.vdqn_softmax_rows(_ptr(x32), _ptr(enc), B, 64, 3, _stream()), "vdqn_softmax_rows")
        return enc[:, :3].clone(), y32.view(B, 64)[:, :3].clone()


def _torchvision_order(keys):
    """Order trunk keys as torchvision's modules register them (conv1, bn1.{weight,bias,running_mean,running_var}, then
    per block conv1, bn1, conv2, bn2, downsample.0, downsample.1)."""
    def rank(k):
        parts = k.split(".")
        i = int(parts[1])
        if i in (0, 1):
            return (i, 0, 0, _leaf(parts[2:]))
        blk = int(parts[2])
        sub = {"conv1": 0, "bn1": 1, "conv2": 2, "bn2": 3, "downsample": 4}[parts[3]]
        if parts[3] == "downsample":
            return (i, blk, sub + int(parts[4]), _leaf(parts[5:]))
        return (i, blk, sub, _leaf(parts[4:]))
    return sorted(keys, key=rank)


def _leaf(parts):
    return {"weight": 0, "bias": 1, "running_mean": 2, "running_var": 3}[parts[-1]]


def label_inverse_actions(feather_path: str, model_path: str, out_path=None, batch_size: int = 64, dtype=None, device=None, log=print):
    """dataset/process_episodes_real.py:84-95,164-179: label every (before_image, after_image) row of the data frame with
    ``model(be, ae)[1].argmax(1)`` and store it as the ``inverse_actions`` column.  Frames are decoded with the loader's
    Resize(224)+CenterCrop(224) (``dataset.resize_center_crop_u8``) and normalised inside the GPU input kernel."""
    import numpy as np
    import pandas as pd
    from PIL import Image

    from .dataset import resize_center_crop_u8
    df = pd.read_feather(feather_path)
    model = InverseActionModel(dtype=dtype, device=device, max_batch=batch_size)
    model.load_state_dict(torch.load(model_path, map_location="cpu"), strict=True)
    model.eval()
    acts = []
    for lo in range(0, len(df), batch_size):
        rows = df.iloc[lo:lo + batch_size]
        be = torch.from_numpy(np.stack([resize_center_crop_u8(Image.open(p)) for p in rows["before_image"]]))
        ae = torch.from_numpy(np.stack([resize_center_crop_u8(Image.open(p)) for p in rows["after_image"]]))
        _, y = model(be.to(model.engine.device), ae.to(model.engine.device))
        acts.append(y.argmax(dim=1).cpu())
        if (lo // batch_size) % 50 == 0:
            log(f"inverse labelling: {lo + len(rows)}/{len(df)}")
    df["inverse_actions"] = torch.cat(acts).numpy()
    df.reset_index(drop=True, inplace=True)
    df.to_feather(out_path or feather_path)
    return df


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 3:
        raise SystemExit("usage: python -m video_dqn_amd.inverse_model <data.feather> <inverse_model.torch> [out.feather]")
    label_inverse_actions(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
