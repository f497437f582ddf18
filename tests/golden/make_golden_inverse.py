#!/usr/bin/env python
"""Golden vectors for the inverse-action model, produced by the REFERENCE class
``archs/inverse_action2.py:45-100`` imported as-is (build container only).  Its module-level imports that are absent
from this image (absl, torch.utils.tensorboard, torchvision) are satisfied by empty stand-ins created in a temp dir —
the class itself only needs ``torchvision.models.resnet18`` (the restated topology of oracle/ref_cpu.py, as for the
other goldens).  Pinned: the state_dict key list/shapes (G8) and the eval-mode outputs ``(encoding, y)`` and the action
labels ``y.argmax(1)`` the data pipeline derives from them (dataset/process_episodes_real.py:172-175) (G9).

Usage:  python tests/golden/make_golden_inverse.py      (writes tests/golden/golden_inverse.npz)
"""
import os
import sys
import tempfile

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from video_dqn_amd import synth  # noqa: E402


def import_reference():
    shim = tempfile.mkdtemp(prefix="invshim_")
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
    os.makedirs(os.path.join(shim, "tensorboard"))
    open(os.path.join(shim, "tensorboard", "__init__.py"), "w").write("")
    sys.path.insert(0, shim)
    sys.path.insert(1, REF)
    # torch.utils.tensorboard imports `tensorboard` lazily and checks its version: give it a stand-in module object
    import types
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    import archs.inverse_action2 as inv
    assert inv.model.__module__ == "archs.inverse_action2"
    return inv.model


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefModel = import_reference()
    m = RefModel()
    sd_ref = m.state_dict()
    out = {"g8_keys": np.array(list(sd_ref.keys())), "g8_shapes": np.array([str(tuple(v.shape)) for v in sd_ref.values()])}
    sd = synth.make_inverse_state_dict(21)
    m.load_state_dict(sd, strict=True)
    m.eval()  # dataset/process_episodes_real.py:94
    for B in (1, 4):
        be = synth.normalise_frames(synth.make_frames_uint8(50 + B, "be", B, 1, structured=True))
        ae = synth.normalise_frames(synth.make_frames_uint8(50 + B, "ae", B, 1, structured=True))
        with torch.no_grad():
            enc, y = m(be, ae)
        out[f"g9_enc_B{B}"] = enc.numpy()
        out[f"g9_y_B{B}"] = y.numpy()
        out[f"g9_act_B{B}"] = y.argmax(dim=1).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_inverse.npz"), **out)
    print("wrote golden_inverse.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
