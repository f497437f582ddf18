"""``HabitatDQNMultiAction`` with the reference's Python surface, computing on the HIP engine.

Mirrors ``archs/HabitatDQNMultiAction.py:8-54`` and the factory/loader of ``train_q_network.py:36-57``:

  * same constructor arguments, same module tree (``resnet``, ``features``, ``top``) so ``state_dict()`` has
    the reference's 250 keys in the reference's order (``features.N.*`` alias ``resnet.*``), and
    ``load_state_dict(strict=True)`` accepts reference checkpoints unchanged;
  * ``forward(inp)`` takes ``float[B,3,224,224]`` / ``float[B,F,3,224,224]`` (normalised, as the reference's
    loader produces) and returns ``float32[B,5,A]``; raises ``Exception("bad shape")`` on a frame-count
    mismatch (:47-48).  Extension: a uint8 ``[B,(F,)224,224,3]`` tensor is normalised on the GPU inside the
    input-packing kernel (``util/torch.py:26-36`` semantics);
  * ``set_train()`` / ``eval()`` / ``train()`` set flags: in ``extra_capacity`` every BatchNorm layer runs on its
    running statistics in both modes (:37-40); in ``basic`` a forward in train mode normalises with batch statistics
    per frame slot and updates ``running_mean/var`` and ``num_batches_tracked`` like torch's BatchNorm2d.

The module's parameters and BatchNorm buffers are *views* into the engine's flat device arrays, so
``load_state_dict`` writes straight into what the kernels read.  The fast training path is the engine's fused update
(``video_dqn_amd.engine.TDStepper``: one 2B-frame online pass, early Adam, no autograd).  The module is ALSO
differentiable for ``extra_capacity`` (the shipped configuration): with grad mode on, ``forward`` runs as a
``torch.autograd.Function`` whose backward is the engine's staged backward of that one call
(``vdqn_net_backward_begin`` + ``vdqn_net_backward_stage``), parameters have ``requires_grad=True`` and receive
``.grad``, so the reference's own loop — ``model(before)``, ``target_net(after)``, ``model(after)``,
``loss.backward()``, ``optim.Adam(model.parameters()).step()`` (train_q_network.py:124,131,140-142,226-227) — runs
unmodified on the HIP kernels (three separate passes and torch's Adam: slower than ``TDStepper``, same numbers).
No arithmetic happens on the CPU.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .engine import NetEngine


class _QFunction(torch.autograd.Function):
    """`model(x)` for autograd: forward = vdqn_net_forward into a workspace this call owns, backward = the engine's staged
    backward of that call from dL/dQ.  `params` are the module's trainable Parameters (slot order); their gradients are slices
    of one flat buffer."""

    @staticmethod
    def forward(ctx, module, frames, src_kind, n_samples, *params):
        eng = module.engine
        packed = module._packed_with_dgrad()
        q, acts = eng.forward_saved(frames, src_kind, n_samples, packed)
        ctx.module, ctx.acts, ctx.packed, ctx.n = module, acts, packed, n_samples
        return q

    @staticmethod
    def backward(ctx, gq):
        module, eng = ctx.module, ctx.module.engine
        flat = eng.backward_from_dq(ctx.acts, ctx.packed, gq, ctx.n)
        ctx.acts = ctx.packed = None
        grads = tuple(flat[s.offset:s.offset + s.numel].view(s.shape) for s in module._trainable_slots)
        return (None, None, None, None) + grads


class _Holder(nn.Module):
    """Parameter/buffer container that only contributes names to the state_dict (no forward)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("structure-only module: HabitatDQNMultiAction.forward runs on the HIP engine")


def _conv(holder_params, name, has_bias=False):
    m = _Holder()
    m.weight = holder_params(name + ".weight")
    m.bias = holder_params(name + ".bias") if has_bias else None
    return m


class _Stateless(_Holder):
    pass


def _init_like_reference(engine: NetEngine, seed=None):
    """Default initialisation when no pretrained file is given: torchvision's resnet init (kaiming_normal
    fan_out for convs, BatchNorm weight 1 / bias 0, running stats 0 / 1) and PyTorch's default Conv2d/Linear
    init for the head, drawn from the torch CPU generator (seeded by the trainer like train_q_network.py:86)."""
    g = None
    if seed is not None:
        g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, s in engine.slots.items():
            v = engine.view(name)
            if s.kind == 2:
                v.zero_()
            elif s.kind == 3:
                v.fill_(1.0)
            elif name.startswith("resnet.") and len(s.shape) == 4:
                fan_out = s.shape[0] * s.shape[2] * s.shape[3]
                v.copy_(torch.randn(s.shape, generator=g) * (2.0 / fan_out) ** 0.5)
            elif name.startswith("resnet.") and (".bn" in name or ".downsample.1." in name):
                v.fill_(1.0 if name.endswith("weight") else 0.0)
            else:  # resnet.fc, features.8, top.*: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                wname = name.rsplit(".", 1)[0] + ".weight"
                ws = engine.slots[wname].shape
                fan_in = ws[1] * (ws[2] * ws[3] if len(ws) == 4 else 1)
                bound = 1.0 / fan_in ** 0.5
                v.copy_((torch.rand(s.shape, generator=g) * 2 - 1) * bound)
    engine.mark_dirty()


def load_torchvision_resnet18(engine: NetEngine, sd, source: str = "state_dict") -> int:
    """`models.resnet18(pretrained=True)` (archs/HabitatDQNMultiAction.py:11) from a file: copy a torchvision-keyed ResNet-18
    state_dict ('conv1.weight', 'layer1.0.bn1.running_mean', ..., 'fc.bias') into the engine's `resnet.*` tensors.  Strict on
    everything the network computes with, as torchvision's own load_state_dict is: EVERY convolution and BatchNorm tensor of the
    trunk (20 convolutions, 20 BatchNorm layers with their running statistics) must be present with its exact shape; a file that
    covers only part of the trunk, or a tensor of the wrong shape (a ResNet-34 / a different width), is an error and nothing is
    copied.  The classifier `fc` is never on the path (archs/HabitatDQNMultiAction.py:30 cuts it off; it only rides along in the
    checkpoint): an ImageNet file's [1000, 512] fc is copied, a differently sized one — the 365-way fc of a Places365 ResNet-18,
    also wrapped as {'state_dict': ..., 'module.' prefixes} — is skipped with a warning and the engine keeps its own.
    `num_batches_tracked` entries are accepted and ignored (eval-mode statistics do not use them); any other unknown key is an
    error.  Returns the number of tensors copied."""
    if hasattr(sd, "state_dict"):
        sd = sd.state_dict()
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]  # (Places365 checkpoints wrap the tensors this way)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    want = {name[len("resnet."):]: s for name, s in engine.slots.items() if name.startswith("resnet.")}
    skip_fc = [k for k in ("fc.weight", "fc.bias") if k in sd and tuple(sd[k].shape) != tuple(want[k].shape)]
    if skip_fc or any(k not in sd for k in ("fc.weight", "fc.bias")):
        import warnings
        warnings.warn(f"{source}: the classifier fc ({[tuple(sd[k].shape) for k in skip_fc] or 'absent'}) does not match resnet.fc [1000, 512]; "
                      "it is not on the Q-network's path and is left at the engine's own values")
        want = {k: v for k, v in want.items() if not k.startswith("fc.")}
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
    missing = sorted(k for k in want if k not in sd)
    unknown = sorted(k for k in sd if k not in want and not k.endswith("num_batches_tracked"))
    wrong = sorted(f"{k}: {tuple(sd[k].shape)} != {tuple(want[k].shape)}" for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k].shape))
    if missing or unknown or wrong:
        raise ValueError(f"{source} is not a torchvision ResNet-18 state_dict: {len(missing)} trunk tensors missing "
                         f"(e.g. {missing[:3]}), {len(unknown)} unknown keys (e.g. {unknown[:3]}), {len(wrong)} wrong shapes (e.g. {wrong[:3]})")
    with torch.no_grad():
        for k in want:
            engine.view("resnet." + k).copy_(sd[k].to(torch.float32))
    engine.mark_dirty()
    return len(want)


class HabitatDQNMultiAction(nn.Module):
    def __init__(self, action_dim, num_classes=5, extra_capacity=False, panorama=True, num_frames=None,
                 dtype=None, device=None, max_batch=64, pretrained_weights=None, deterministic=None):
        super().__init__()
        self.extra_capacity = extra_capacity
        self.num_classes = num_classes
        self.action_dim = action_dim
        self.panorama = panorama
        if num_frames is None:  # archs/HabitatDQNMultiAction.py:16-19
            num_frames = 4 if panorama else 1
        self.num_frames = num_frames
        dtype = dtype or os.environ.get("VDQN_DTYPE", "bf16")
        if extra_capacity:
            print("Model loading with extra_capacity")  # :28
        self.engine = NetEngine(action_dim, num_classes, num_frames, extra_capacity, dtype, max_batch, device, deterministic)
        self._build_tree()
        _init_like_reference(self.engine)
        # models.resnet18(pretrained=True) (:11): the ImageNet weights come from a file (config key PRETRAINED_WEIGHTS, or the
        # VDQN_RESNET18_WEIGHTS environment variable) — without one the trunk keeps its random initialisation, and
        # `pretrained_loaded` stays False so the trainer can say so
        pre = pretrained_weights or os.environ.get("VDQN_RESNET18_WEIGHTS")
        self.pretrained_loaded = False
        if pre:
            load_torchvision_resnet18(self.engine, torch.load(pre, map_location="cpu"), source=str(pre))
            self.pretrained_loaded = True

    # ---- structure ----------------------------------------------------------------------------------
    def _build_tree(self):
        eng = self.engine
        cache = {}
        bn_count = [0]

        def P(name):
            # requires_grad as in the reference (every nn.Parameter, the never-used resnet.fc included: it simply never
            # receives a .grad, train_q_network.py:124); the Parameters share storage AND version counter with eng.params
            if name not in cache:
                cache[name] = nn.Parameter(eng.view(name), requires_grad=True)
            return cache[name]

        def bn(name):
            m = _Holder()
            m.weight, m.bias = P(name + ".weight"), P(name + ".bias")
            m.register_buffer("running_mean", eng.view(name + ".running_mean"))
            m.register_buffer("running_var", eng.view(name + ".running_var"))
            m.register_buffer("num_batches_tracked", eng.num_batches_tracked[bn_count[0]])  # 0-dim view
            bn_count[0] += 1
            return m

        resnet = _Holder()
        resnet.conv1 = _conv(P, "resnet.conv1")
        resnet.bn1 = bn("resnet.bn1")
        resnet.relu = _Stateless()
        resnet.maxpool = _Stateless()
        inpl = 64
        for li, planes in enumerate((64, 128, 256, 512), start=1):
            blocks = []
            for bi in range(2):
                p = f"resnet.layer{li}.{bi}"
                b = _Holder()
                b.conv1 = _conv(P, p + ".conv1")
                b.bn1 = bn(p + ".bn1")
                b.relu = _Stateless()
                b.conv2 = _conv(P, p + ".conv2")
                b.bn2 = bn(p + ".bn2")
                if (li > 1 and bi == 0) or inpl != planes:
                    b.downsample = nn.Sequential(_conv(P, p + ".downsample.0"), bn(p + ".downsample.1"))
                else:
                    b.downsample = None
                inpl = planes
                blocks.append(b)
            setattr(resnet, f"layer{li}", nn.Sequential(*blocks))
        resnet.avgpool = _Stateless()
        resnet.fc = _conv(P, "resnet.fc", has_bias=True)
        self.resnet = resnet
        # what torch.autograd differentiates `forward` with respect to: the trainable slots, in the engine's flat order
        self._trainable_slots = [s for s in eng.slots.values() if s.kind == 0]
        self._trainable_params = None  # filled after the tree is complete (below)
        self._cache = cache
        if self.extra_capacity:  # archs/HabitatDQNMultiAction.py:30-31
            self.features = nn.Sequential(*list(self.resnet.children())[:-2], _conv(P, "features.8", has_bias=True),
                                          _Stateless(), _Stateless())
            self.top = nn.Sequential(_conv(P, "top.0", True), _Stateless(), _conv(P, "top.2", True), _Stateless(),
                                     _conv(P, "top.4", True))
        else:  # :33-34
            self.features = nn.Sequential(*list(self.resnet.children())[:-1])
            self.top = _conv(P, "top", True)
        object.__setattr__(self, "_trainable_params", [cache[s.name] for s in self._trainable_slots])
        del self._cache
        self._packed_grad = None      # packed weights incl. the data-gradient operands, for differentiable forwards
        self._packed_grad_key = None  # (engine version, version counter of the flat parameter array) they were packed from

    def _packed_with_dgrad(self):
        """The packed weights (BatchNorm folded, forward AND data-gradient operands) of the CURRENT parameters.  In-place
        updates by torch optimisers are seen through the version counter the Parameters share with the flat array; a new
        buffer is used whenever the parameters changed, so a graph that still holds the old one differentiates the weights
        its forward ran with."""
        eng = self.engine
        key = eng.version_key()
        if self._packed_grad is None or self._packed_grad_key != key:
            with torch.cuda.device(eng.device):
                self._packed_grad = torch.empty(eng.packed_bytes, dtype=torch.uint8, device=eng.device)
                eng.pack_weights(self._packed_grad, with_dgrad=True)
            self._packed_grad_key = key
        return self._packed_grad

    # ---- reference surface ----------------------------------------------------------------------------
    def set_train(self):  # :37-40
        self.train()
        if self.extra_capacity:
            self.resnet.eval()

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.engine.mark_dirty()
        return r

    def _apply(self, fn, recurse=True):
        # parameters are views into the engine's flat arrays; moving them would break the aliasing
        probe = fn(torch.empty(0, device=self.engine.device))
        if probe.device != self.engine.device or probe.dtype != torch.float32:
            raise RuntimeError("HabitatDQNMultiAction lives on its engine's device in f32 master precision; "
                               "construct it with device=... instead of calling .to()/.half()")
        return self

    def forward(self, inp):  # :44-54
        eng = self.engine
        if inp.dtype == torch.uint8:  # raw frames [B,(F,)224,224,3]
            if self.num_frames == 1 and inp.dim() == 4:
                inp = inp.unsqueeze(1)
            if inp.shape[1] != self.num_frames:
                raise Exception("bad shape")
            src_kind = 0
        else:
            if self.num_frames == 1 and len(inp.shape) == 4:
                inp = inp.unsqueeze(1)
            if inp.shape[1] != self.num_frames:
                raise Exception("bad shape")
            inp = inp.float()
            src_kind = 1
        if inp.requires_grad:
            raise NotImplementedError("HabitatDQNMultiAction on the HIP engine differentiates with respect to its parameters only "
                                      "(the reference never asks for the gradient of a frame): detach the input")
        b = inp.shape[0]
        inp = inp.to(eng.device).contiguous()
        if b > eng.max_batch:
            outs = [self.forward(inp[i:i + eng.max_batch]) for i in range(0, b, eng.max_batch)]
            return torch.cat(outs, 0)
        if self.training and not self.extra_capacity:  # the ResNet's BatchNorm layers are in train mode (:37-40)
            q = eng.forward_train(inp, src_kind, b)  # (no autograd graph: 'basic' trains through TDStepper)
        elif self.extra_capacity and torch.is_grad_enabled() and any(p.requires_grad for p in self._trainable_params):
            q = _QFunction.apply(self, inp, src_kind, b, *self._trainable_params)
        else:
            q = eng.forward(inp, src_kind, b)
        return q.view((-1, self.num_classes, self.action_dim))


def build_model(config, max_batch=None):
    """train_q_network.py:36-47 (config.device selects the GPU; dtype from config.COMPUTE_DTYPE)."""
    actions = 1 if (config.VALUE_LEARNING or config.ONE_ACTION) else 3
    nf = getattr(config, "NUM_FRAMES", 0) or None
    return HabitatDQNMultiAction(actions, 5, extra_capacity=(config.ARCHITECTURE == "extra_capacity"),
                                 panorama=(config.PANORAMA or config.PREVIOUS_IMAGES), num_frames=nf,
                                 dtype=getattr(config, "COMPUTE_DTYPE", None), device=config.device,
                                 max_batch=max_batch or max(64, 2 * getattr(config, "BATCH_SIZE", 16)),
                                 pretrained_weights=(getattr(config, "PRETRAINED_WEIGHTS", "") or None),
                                 deterministic=(True if getattr(config, "DETERMINISTIC", False) else None))


def load_model_number(config, number, model_loc=None):
    """train_q_network.py:50-57."""
    model = build_model(config)
    if model_loc is None:
        model_loc = f"{config.folder}/models/sample{number}.torch"
    snapshot = torch.load(model_loc, map_location=config.device)
    print(f"Loading model from: {model_loc}")
    model.load_state_dict(snapshot["model_state_dict"])
    return model
