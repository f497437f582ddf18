#!/usr/bin/env python
"""Golden vectors for TRAINING the inverse-action model, produced by the reference's own training-script model class
(``train_inverse_model.py:30-83``, imported as-is with stand-ins for absl / tensorboard / torchvision / the Gibson
dataloader module, none of which the class itself uses beyond ``FLAGS.bottleneck_size``) and the statements of its
training loop (:93-112: CrossEntropyLoss, backward, Adam step).  Dropout2d(0.5) draws from torch's RNG: the masks it
drew are captured with a forward hook and stored, so the GPU path and the oracle replay exactly the same masks.

Usage:  python tests/golden/make_golden_inverse_train.py      (writes tests/golden/golden_inverse_train.npz)
"""
import os
import sys
import tempfile
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import sample_idx  # noqa: E402
from video_dqn_amd import synth  # noqa: E402


def import_reference():
    shim = tempfile.mkdtemp(prefix="invtshim_")
    os.makedirs(os.path.join(shim, "torchvision"))
    open(os.path.join(shim, "torchvision", "__init__.py"), "w").write("from . import models, transforms\n")
    open(os.path.join(shim, "torchvision", "models.py"), "w").write("from oracle.ref_cpu import resnet18\n")
    open(os.path.join(shim, "torchvision", "transforms.py"), "w").write("")
    os.makedirs(os.path.join(shim, "absl"))
    open(os.path.join(shim, "absl", "__init__.py"), "w").write("from . import app, flags\n")
    open(os.path.join(shim, "absl", "app.py"), "w").write("def run(main):\n    raise SystemExit('stub')\n")
    open(os.path.join(shim, "absl", "flags.py"), "w").write(
        "class _F:\n    pass\nFLAGS = _F()\n"
        "def _d(name, default, doc):\n    setattr(FLAGS, name, default)\n"
        "DEFINE_integer = DEFINE_float = DEFINE_string = _d\n")
    sys.path.insert(0, shim)
    sys.path.insert(1, REF)
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    gib = types.ModuleType("dataloaders.gibson")  # the training script imports its dataset class at module level
    gib.GibsonDatasetPair = object
    pkg = types.ModuleType("dataloaders")
    pkg.__path__ = []
    sys.modules["dataloaders"] = pkg
    sys.modules["dataloaders.gibson"] = gib
    import train_inverse_model as tim
    assert tim.model.__module__ == "train_inverse_model"
    return tim.model


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefModel = import_reference()
    m = RefModel()
    m.load_state_dict(synth.make_inverse_state_dict(21), strict=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=0.0)  # train_inverse_model.py:190 with the flag defaults
    masks = []
    m.dropout1.register_forward_hook(lambda mod, inp, outp: masks.append(((outp != 0) | (inp[0] == 0)).float()))
    out = {}
    B = 6
    for step in (1, 2):
        be = synth.normalise_frames(synth.make_frames_uint8(70 + step, "be", B, 1, structured=True))
        ae = synth.normalise_frames(synth.make_frames_uint8(70 + step, "ae", B, 1, structured=True))
        act = torch.from_numpy(synth.randint(70 + step, "act", (B,), 3))
        m.train()  # :88
        opt.zero_grad()  # :96
        y = m(be, ae)  # :99
        loss = torch.nn.CrossEntropyLoss()(y, act)  # :102-104
        loss.backward()  # :111
        k = f"g10_s{step}"
        out[f"{k}_mask"] = masks[-1].numpy()
        out[f"{k}_loss"] = np.array(loss.item(), dtype=np.float64)
        out[f"{k}_y"] = y.detach().numpy()
        for n, p in m.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().flatten()
            out[f"{k}_gnorm_{n}"] = np.array(g.double().norm().item())
            out[f"{k}_gabsmax_{n}"] = np.array(g.abs().max().item())
            out[f"{k}_gsamp_{n}"] = g[sample_idx(n, g.numel())].numpy()
        opt.step()  # :112
        for n, p in m.named_parameters():
            if p.requires_grad:
                out[f"{k}_psamp_{n}"] = p.detach().flatten()[sample_idx(n, p.numel())].numpy()
    out["g10_trainable"] = np.array([n for n, p in m.named_parameters() if p.requires_grad])
    np.savez_compressed(os.path.join(HERE, "golden_inverse_train.npz"), **out)
    print("wrote golden_inverse_train.npz:", len(out), "arrays; trainable:", list(out["g10_trainable"]))


if __name__ == "__main__":
    main()
