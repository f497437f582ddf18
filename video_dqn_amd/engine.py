"""Host-side driver of the HIP engine (libvdqn.so): owns the torch-allocated device buffers and calls
the C ABI.  PyTorch is used for device memory and streams only — all arithmetic runs in the HIP library.

Mirrors the pieces of ``train_q_network.py`` that touch the device:
  * model / target_net / Adam construction       (:119-124)
  * target sync                                   (:121,208,215-216)
  * zero_grad / process_batch / backward / step   (:222-227)
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib

LOSS_KINDS = {"l2": 0, "huber": 1}  # vdqn_td_args.loss_kind

DTYPES = {"f32": _lib.VDQN_F32, "fp32": _lib.VDQN_F32, "float32": _lib.VDQN_F32,
          "bf16": _lib.VDQN_BF16, "bfloat16": _lib.VDQN_BF16}


# VDQN_EARLY_ADAM=0: `TDStepper.step` runs the whole optimiser update behind the backward pass (one launch)
_EARLY_ADAM = os.environ.get("VDQN_EARLY_ADAM", "1") != "0"
# VDQN_DIST_EARLY_ADAM=1: under a gradient exchange, stage 0 / 1 are updated behind their own bucket (TDStepper.allreduce_wait)
_DIST_EARLY_ADAM = os.environ.get("VDQN_DIST_EARLY_ADAM", "0") in ("1", "2", "3")
# diagnostic values (one rank only — they are wrong with more): 2 = the same without waiting for the bucket's collective (is the wait
# the cost?), 3 = waiting for it, but without the Adam launch (is the extra kernel the cost?)
_DIST_EARLY_ADAM_MODE = os.environ.get("VDQN_DIST_EARLY_ADAM", "0")



def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.VdqnError("video_dqn_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


@dataclass
class ParamSlot:
    name: str
    offset: int
    numel: int
    shape: tuple
    kind: int      # 0 trainable, 1 frozen (resnet.fc), 2 running_mean, 3 running_var
    param_id: int
    stage: int


class NetEngine:
    """One HabitatDQNMultiAction instance on the device: flat f32 master parameters + BN statistics."""

    def __init__(self, action_dim=3, num_classes=5, num_frames=1, extra_capacity=True, dtype="bf16",
                 max_batch=512, device=None, deterministic=None):
        self.lib = _lib.load()
        # device="cpu" gives a storage-only instance (state_dict / checkpoint plumbing); every compute entry
        # point still requires the GPU and raises otherwise — there is no CPU arithmetic in this package.
        self.storage_only = device is not None and torch.device(device).type == "cpu"
        self.device = torch.device("cpu") if self.storage_only else require_gpu(device)
        self.extra_capacity = bool(extra_capacity)
        self.action_dim, self.num_classes, self.num_frames = action_dim, num_classes, num_frames
        self.dtype_name = "bf16" if DTYPES[dtype] == _lib.VDQN_BF16 else "f32"
        # run-to-run bit-identical updates (the reference pins cudnn.deterministic, train_q_network.py:88-89): config key
        # DETERMINISTIC / this argument, or VDQN_DETERMINISTIC=1 in the environment
        if deterministic is None:
            deterministic = os.environ.get("VDQN_DETERMINISTIC", "0") == "1"
        self.deterministic = bool(deterministic)
        self.cfg = _lib.NetConfig(action_dim, num_classes, num_frames, int(self.extra_capacity), DTYPES[dtype], max_batch,
                                  int(self.deterministic))
        h = C.c_void_p()
        _lib.check(self.lib.vdqn_net_create(C.byref(self.cfg), C.byref(h)), "vdqn_net_create")
        self.handle = h
        self.max_batch = max_batch
        self.slots: "OrderedDict[str, ParamSlot]" = OrderedDict()
        info = _lib.ParamInfo()
        for i in range(self.lib.vdqn_net_num_params(h)):
            _lib.check(self.lib.vdqn_net_param_info(h, i, C.byref(info)), "vdqn_net_param_info")
            name = info.name.decode()
            self.slots[name] = ParamSlot(name, info.offset, info.numel, tuple(info.shape[:info.ndim]), info.kind,
                                         info.param_id, info.stage)
        self.params_numel = self.lib.vdqn_net_params_numel(h)
        self.trainable_numel = self.lib.vdqn_net_trainable_numel(h)
        self.bnstats_numel = self.lib.vdqn_net_bnstats_numel(h)
        self.packed_bytes = self.lib.vdqn_net_packed_bytes(h)
        self.params = torch.zeros(self.params_numel, dtype=torch.float32, device=self.device)
        self.bnstats = torch.zeros(self.bnstats_numel, dtype=torch.float32, device=self.device)
        # BatchNorm.num_batches_tracked of the 20 BatchNorm layers (only ever advanced by ARCHITECTURE='basic',
        # whose BatchNorm layers run in train mode)
        self.num_batches_tracked = torch.zeros(20, dtype=torch.long, device=self.device)
        self.packed = None if self.storage_only else torch.zeros(self.packed_bytes, dtype=torch.uint8, device=self.device)
        self._packed_version = None
        self._version = 0
        self._acts: Dict[int, torch.Tensor] = {}

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.vdqn_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- parameter views -------------------------------------------------------------------------
    def view(self, name: str) -> torch.Tensor:
        s = self.slots[name]
        base = self.params if s.kind in (0, 1) else self.bnstats
        return base[s.offset:s.offset + s.numel].view(s.shape)

    def stage_range(self, stage: int):
        b, e = C.c_int64(), C.c_int64()
        _lib.check(self.lib.vdqn_net_stage_range(self.handle, stage, C.byref(b), C.byref(e)), "stage_range")
        return b.value, e.value

    def load_tensors(self, state_dict) -> None:
        """Copy every tensor the engine owns (by its reference state_dict name) from ``state_dict``."""
        with torch.no_grad():
            for name in self.slots:
                self.view(name).copy_(state_dict[name].to(torch.float32))
        self.mark_dirty()

    def mark_dirty(self):
        """Call after the master parameters changed through raw pointers (the Adam kernel, a broadcast)."""
        self._version += 1

    def version_key(self):
        """Identifies the parameter values the packed weights were made from: the explicit counter above plus the autograd
        version counters of the flat arrays, which every in-place torch op on a view of them (the module's Parameters under a
        torch optimiser, load_state_dict) advances."""
        return (self._version, self.params._version, self.bnstats._version)

    # ---- packing + forward -----------------------------------------------------------------------
    def _need_gpu(self):
        if self.storage_only:
            raise _lib.VdqnError("this NetEngine is storage-only (device='cpu'); compute needs the GPU — no CPU fallback")

    def pack_weights(self, packed: Optional[torch.Tensor] = None, with_dgrad: bool = False, params=None, bnstats=None):
        self._need_gpu()
        packed = self.packed if packed is None else packed
        _lib.check(self.lib.vdqn_net_pack_weights(self.handle, _ptr(self.params if params is None else params),
                                                  _ptr(self.bnstats if bnstats is None else bnstats),
                                                  _ptr(packed), int(with_dgrad), _stream()), "vdqn_net_pack_weights")

    def acts_bytes(self, n_samples: int) -> int:
        return self.lib.vdqn_net_acts_bytes(self.handle, n_samples)

    def bwd_bytes(self, n_samples: int) -> int:
        return self.lib.vdqn_net_bwd_bytes(self.handle, n_samples)

    def _acts_for(self, n: int) -> torch.Tensor:
        buf = self._acts.get(n)
        if buf is None:
            self._acts.clear()
            buf = torch.empty(self.acts_bytes(n), dtype=torch.uint8, device=self.device)
            self._acts[n] = buf
        return buf

    def forward(self, frames: torch.Tensor, src_kind: int, n_samples: int) -> torch.Tensor:
        """frames: contiguous device tensor (uint8 NHWC frames if src_kind == 0, f32 NCHW if 1)."""
        self._need_gpu()
        if n_samples > self.max_batch:
            raise _lib.VdqnError(f"batch {n_samples} exceeds max_batch {self.max_batch}")
        with torch.cuda.device(self.device):
            if self._packed_version != self.version_key():
                self.pack_weights()
                self._packed_version = self.version_key()
            q = torch.empty((n_samples, self.num_classes * self.action_dim), dtype=torch.float32, device=self.device)
            acts = self._acts_for(n_samples)
            _lib.check(self.lib.vdqn_net_forward(self.handle, _ptr(self.packed), _ptr(frames), src_kind, n_samples,
                                                 _ptr(acts), _ptr(q), _stream()), "vdqn_net_forward")
        return q

    # ---- one model call with its own backward (torch.autograd over the module, model.py) ----------------------------
    def forward_saved(self, frames: torch.Tensor, src_kind: int, n_samples: int, packed: torch.Tensor):
        """`vdqn_net_forward` into a workspace of its own that the caller keeps for `backward_from_dq`: -> (q, acts).
        `packed` must hold this network's weights packed WITH the data-gradient operands (`pack_weights(.., with_dgrad=True)`)."""
        self._need_gpu()
        if n_samples > self.max_batch:
            raise _lib.VdqnError(f"batch {n_samples} exceeds max_batch {self.max_batch}")
        with torch.cuda.device(self.device):
            q = torch.empty((n_samples, self.num_classes * self.action_dim), dtype=torch.float32, device=self.device)
            acts = torch.empty(self.acts_bytes(n_samples), dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.vdqn_net_forward(self.handle, _ptr(packed), _ptr(frames), src_kind, n_samples, _ptr(acts), _ptr(q),
                                                 _stream()), "vdqn_net_forward")
        return q, acts

    def backward_from_dq(self, acts: torch.Tensor, packed: torch.Tensor, dq: torch.Tensor, n_samples: int) -> torch.Tensor:
        """Backward of one `forward_saved` call from dL/dQ (f32 [n_samples, num_classes*action_dim]): the flat f32 gradient
        over the trainable range, complete on the current stream (loss.backward() through `model(before)`,
        train_q_network.py:131,226).  extra_capacity only."""
        self._need_gpu()
        with torch.cuda.device(self.device):
            bwd = getattr(self, "_bwd_ws", None)
            if bwd is None or bwd[0] != n_samples:
                bwd = self._bwd_ws = (n_samples, torch.empty(self.bwd_bytes(n_samples), dtype=torch.uint8, device=self.device))
            grads = torch.zeros(self.trainable_numel, dtype=torch.float32, device=self.device)
            dq = dq.to(device=self.device, dtype=torch.float32).contiguous()
            a = _lib.StepArgs()
            a.params, a.bnstats, a.packed_online = _ptr(self.params), _ptr(self.bnstats), _ptr(packed)
            a.batch = a.acts_samples = n_samples
            a.acts_online, a.bwd, a.grads = _ptr(acts), _ptr(bwd[1]), _ptr(grads)
            st = _stream()
            _lib.check(self.lib.vdqn_net_backward_begin(self.handle, C.byref(a), _ptr(dq), st), "vdqn_net_backward_begin")
            for stage in range(3):  # stage 2 joins the engine's gradient stream back into `st`
                _lib.check(self.lib.vdqn_net_backward_stage(self.handle, C.byref(a), stage, st), "vdqn_net_backward_stage")
            dq.record_stream(torch.cuda.current_stream())
        return grads

    # ---- SyncBN (ARCHITECTURE='basic' under data parallelism) ----------------------------------------
    def set_bn_sync(self, world_size: int, allreduce=None) -> None:
        """Make the train-mode BatchNorm statistics global over ``world_size`` ranks (N ranks == one big batch, as the
        single-process reference computes them).  ``allreduce(t)`` must SUM-all-reduce the f32 device tensor ``t`` in
        place on the current stream; default: ``torch.distributed.all_reduce``.  world_size <= 1 switches it off."""
        self._need_gpu()
        if world_size <= 1:
            _lib.check(self.lib.vdqn_net_set_bn_sync(self.handle, None, None, 1), "vdqn_net_set_bn_sync")
            self._bn_sync_cb = None
            return
        if allreduce is None:
            import torch.distributed as dist

            def allreduce(t):
                dist.all_reduce(t)
        owners = self._sync_buffers = getattr(self, "_sync_buffers", [])

        def cb(user, buf, count, stream):
            # `buf` lies inside one of the activation workspaces this engine was handed: wrap it as a tensor view
            for ref in owners:
                t = ref()
                if t is None:
                    continue
                off = buf - t.data_ptr()
                if 0 <= off and off + 4 * count <= t.numel():
                    allreduce(t[off:off + 4 * count].view(torch.float32))
                    return
            raise _lib.VdqnError("SyncBN: statistics buffer is not inside a registered activation workspace")
        self._bn_sync_cb = _lib.ALLREDUCE_FN(cb)  # keep the trampoline alive
        _lib.check(self.lib.vdqn_net_set_bn_sync(self.handle, C.cast(self._bn_sync_cb, C.c_void_p), None, int(world_size)),
                   "vdqn_net_set_bn_sync")

    def register_sync_buffer(self, t: torch.Tensor) -> None:
        import weakref
        lst = self._sync_buffers = getattr(self, "_sync_buffers", [])
        lst[:] = [r for r in lst if r() is not None]
        if not any(r() is t for r in lst):
            lst.append(weakref.ref(t))

    def forward_train(self, frames: torch.Tensor, src_kind: int, n_samples: int) -> torch.Tensor:
        """ARCHITECTURE='basic' with the module in train mode: batch statistics per frame slot, running statistics and
        num_batches_tracked updated (one model call, archs/HabitatDQNMultiAction.py:44-54 under model.train())."""
        self._need_gpu()
        if self.extra_capacity:
            raise _lib.VdqnError("forward_train: extra_capacity keeps BatchNorm in eval mode; use forward()")
        if n_samples > self.max_batch:
            raise _lib.VdqnError(f"batch {n_samples} exceeds max_batch {self.max_batch}")
        with torch.cuda.device(self.device):
            q = torch.empty((n_samples, self.num_classes * self.action_dim), dtype=torch.float32, device=self.device)
            acts = self._acts_for(n_samples)
            self.register_sync_buffer(acts)
            self._packed_version = None  # self.packed now holds the un-folded weights
            _lib.check(self.lib.vdqn_net_forward_train(self.handle, _ptr(self.params), _ptr(self.bnstats), _ptr(self.packed),
                                                       _ptr(frames), src_kind, n_samples, _ptr(acts), _ptr(q), _stream()),
                       "vdqn_net_forward_train")
            self.num_batches_tracked += self.num_frames
        return q


class TDStepper:
    """Device state of the training loop: target-network weights, Adam moments, workspaces, and one TD update.

    ``step()`` follows train_q_network.py:213-227: target sync check (before the update), forward x3 (the two
    online passes run as one 2B batch), Double-DQN target + loss, backward, Adam."""

    def __init__(self, net: NetEngine, batch: int, lr: float, gamma: float, clip_rect: bool, linear: bool = False,
                 remove_before_reward: bool = False, train_on_ground_truth: bool = False, value_learning: bool = False,
                 target_update_interval: int = 8000, betas=(0.9, 0.999), eps: float = 1e-8, world_size: int = 1,
                 allreduce=None, loss_kind: str = "l2", allreduce_loss=None, allreduce_wait=None):
        net._need_gpu()
        self.net, self.B = net, batch
        self.lib = net.lib
        if 2 * batch > net.max_batch:
            raise _lib.VdqnError(f"TDStepper(batch={batch}) needs NetEngine(max_batch>={2 * batch})")
        self.lr, self.gamma, self.betas, self.eps = lr, gamma, betas, eps
        self.clip_rect, self.linear, self.rbr = clip_rect, linear, remove_before_reward
        self.gtb, self.value_learning = train_on_ground_truth, value_learning
        if loss_kind not in LOSS_KINDS:
            raise _lib.VdqnError(f"loss_kind must be one of {sorted(LOSS_KINDS)} (reference: 'l2', train_q_network.py:167)")
        self.loss_kind = loss_kind
        self.tui = target_update_interval
        self.world_size = world_size
        self.allreduce = allreduce  # callable(tensor_slice, stage) or None
        # callable(loss) or None: called once per update behind the LAST gradient bucket, on the gradient stream — the exchange
        # sums the ranks' loss shares there, so the compute stream never waits for a 4-byte collective (dist.launch_loss)
        self.allreduce_loss = allreduce_loss
        # callable() or None: orders the current stream behind the bucket launched last (dist.wait_last).  With VDQN_DIST_EARLY_ADAM=1
        # the optimiser update of stage 0 / stage 1 then runs behind ITS bucket on a stream of its own, under the rest of the backward
        # pass, instead of behind the last bucket (off by default: DESIGN.md section 7 has the measurement)
        self.allreduce_wait = allreduce_wait
        self._post_stream = None
        self._post_used = False
        dev = net.device
        nt = net.trainable_numel
        with torch.cuda.device(dev):
            self.packed_online = torch.zeros(net.packed_bytes, dtype=torch.uint8, device=dev)
            self.packed_target = torch.zeros(net.packed_bytes, dtype=torch.uint8, device=dev)
            n_online = batch if self.gtb else 2 * batch
            self.layout_samples = n_online  # what vdqn_net_act_offset must be asked for to find a tensor inside acts_online
            self.acts_online = torch.empty(net.acts_bytes(n_online), dtype=torch.uint8, device=dev)
            self.acts_target = None if self.gtb else torch.empty(net.acts_bytes(batch), dtype=torch.uint8, device=dev)
            net.register_sync_buffer(self.acts_online)
            self.bwd = torch.empty(net.bwd_bytes(batch), dtype=torch.uint8, device=dev)
            self.grads = torch.zeros(nt, dtype=torch.float32, device=dev)
            self.exp_avg = torch.zeros(nt, dtype=torch.float32, device=dev)
            self.exp_avg_sq = torch.zeros(nt, dtype=torch.float32, device=dev)
            self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
            self.q_before = torch.zeros((batch, net.num_classes * net.action_dim), dtype=torch.float32, device=dev)
            self._ones = torch.ones((batch, net.num_classes), dtype=torch.float32, device=dev)
        self.adam_step = 0
        self.sample_number = 0
        self._grad_stream = None  # torch view of the engine's side stream (vdqn_net_grad_stream)
        self._adam_done = []
        self._packed_bufs, self._ahead = [None, None], None
        self.stage_ranges = [net.stage_range(s) for s in range(3)]
        self.sync_target()

    def sync_target(self):
        """target_net.load_state_dict(model.state_dict()) (train_q_network.py:121,208,216): the target network
        only ever runs forward, so its state is the packed (BN-folded) copy of the current online weights."""
        with torch.cuda.device(self.net.device):
            self.net.pack_weights(self.packed_target, with_dgrad=False)

    def _args(self, before, after, src_kind, act, rew, term, valid, gt) -> _lib.StepArgs:
        n = self.net
        a = _lib.StepArgs()
        a.params, a.bnstats = _ptr(n.params), _ptr(n.bnstats)
        a.packed_online, a.packed_target = _ptr(self.packed_online), _ptr(self.packed_target)
        a.before, a.after, a.src_kind, a.batch = _ptr(before), _ptr(after), src_kind, self.B
        a.act, a.rew, a.term, a.valid, a.gt = _ptr(act), _ptr(rew), _ptr(term), _ptr(valid), _ptr(gt)
        a.gamma = self.gamma
        a.inv_count = 1.0 / (n.num_classes * self.B * self.world_size)
        a.clip_rect, a.linear, a.use_valid = int(self.clip_rect), int(self.linear), int(self.rbr)
        a.train_on_ground_truth, a.value_learning = int(self.gtb), int(self.value_learning)
        a.acts_online, a.acts_target, a.bwd = _ptr(self.acts_online), _ptr(self.acts_target), _ptr(self.bwd)
        a.grads, a.loss, a.q_before = _ptr(self.grads), _ptr(self.loss), _ptr(self.q_before)
        a.loss_kind = LOSS_KINDS[self.loss_kind]
        a.packed_frames = None
        return a

    # ---- frames packed one update ahead (vdqn_step_args.packed_frames) -----------------------------------------------------
    def _dist_early_ok(self) -> bool:
        """Per-bucket Adam under an exchange (VDQN_DIST_EARLY_ADAM) is taken only where it is right: modes 2 / 3 are one-rank
        diagnostics (mode 2 would update with gradients that have not been reduced) and raise with more ranks; mode 1 needs a
        backend whose Work.wait() orders the STREAM (RCCL) — gloo's blocks the host thread, which would serialise the enqueue of the
        remaining backward stages — and falls back to the finish-time update elsewhere."""
        ok = getattr(self, "_dist_early_checked", None)
        if ok is None:
            ok = True
            if self.world_size > 1:
                if _DIST_EARLY_ADAM_MODE in ("2", "3"):
                    raise RuntimeError("VDQN_DIST_EARLY_ADAM=2/3 are one-rank diagnostics: with world_size > 1 they apply unreduced gradients")
                try:
                    import torch.distributed as dist
                    ok = not dist.is_initialized() or dist.get_backend() == "nccl"
                except Exception:  # noqa: BLE001 - a caller-supplied exchange without torch.distributed
                    ok = True
            self._dist_early_checked = ok
        return ok

    @staticmethod
    def _frames_key(before, after, src_kind):
        # identity AND content version: a loop that refills fixed staging buffers in place (copy_ advances `_version`) announces
        # a different key than the one that arrives, so the stale pack is discarded and the call packs its own frames
        return (before.data_ptr(), None if after is None else after.data_ptr(), int(src_kind), tuple(before.shape), before.dtype,
                before._version, None if after is None else after._version)

    def _packed_buffer(self, slot: int) -> torch.Tensor:
        if self._packed_bufs[slot] is None:
            n = self.net
            esz = 2 if n.dtype_name == "bf16" else 4
            nbytes = 2 * self.B * n.num_frames * 115 * 115 * 16 * esz
            self._packed_bufs[slot] = torch.empty(nbytes, dtype=torch.uint8, device=n.device)
        return self._packed_bufs[slot]

    def _pack_ahead(self, next_frames, slot: int):
        """Queue vdqn_pack_input for the NEXT update's frames on the engine's gradient stream — behind this update's target pass,
        i.e. under its head and backward pass — into packed-frame buffer `slot` (the other one is being read by this update)."""
        nb, na, nk = next_frames
        n = self.net
        buf = self._packed_buffer(slot)
        nf = self.B * n.num_frames
        half = buf.numel() // 2
        dt = _lib.VDQN_BF16 if n.dtype_name == "bf16" else _lib.VDQN_F32
        main = torch.cuda.current_stream()
        with self._grad_stream_ctx():
            # the announced tensors may have been produced on the caller's stream a moment ago (a host-to-device copy, a gather)
            torch.cuda.current_stream().wait_stream(main)
            _lib.check(self.lib.vdqn_pack_input(_ptr(nb), int(nk), buf.data_ptr(), nf, dt, _stream()), "vdqn_pack_input")
            if na is not None and not self.gtb:
                _lib.check(self.lib.vdqn_pack_input(_ptr(na), int(nk), buf.data_ptr() + half, nf, dt, _stream()), "vdqn_pack_input")
        self._ahead = (self._frames_key(nb, na, nk), slot, (nb, na))  # (the tensors are kept alive until they are consumed)

    def forward_backward(self, before, after, src_kind, act, rew, term, valid=None, gt=None, early_adam: bool = False, next_frames=None):
        """Everything of one update up to (and including) the gradient all-reduce; no optimiser step.

        early_adam (only `step` passes it: single process, no exchange, eval-mode BatchNorm): the optimiser update of stage 0 and stage 1 is
        queued on the engine's gradient stream right behind that stage's gradient unpack, so it runs under the remaining data /
        weight gradients instead of behind them (nothing later in the update reads those master parameters: the kernels work on
        the packed copies); `optimizer_step` then only covers what is left.  Same arithmetic, same results."""
        n = self.net
        self._adam_done = []
        keep = (before, after, act, rew, term, valid, gt)  # keep inputs alive until the launches are queued
        with torch.cuda.device(n.device):
            a = self._args(before, after, src_kind, act, rew, term, valid if valid is not None else self._ones, gt)
            st = _stream()
            ahead, self._ahead = self._ahead, None
            slot = None
            if ahead is not None and ahead[0] == self._frames_key(before, after, src_kind):
                slot = ahead[1]  # these frames were packed during the previous update (ordered in front of `st` by its last stage)
                a.packed_frames = self._packed_buffer(slot).data_ptr()
            _lib.check(self.lib.vdqn_net_td_forward(n.handle, C.byref(a), st), "vdqn_net_td_forward")
            if next_frames is not None:
                # next_frames that ALIAS this call's tensors (same storage) are packed ahead only under the caller's explicit promise
                # that the content stays as it is — a fourth element True: a loop that replays one resident minibatch.  Without the
                # promise an aliased announcement is ignored (the next call packs its own frames, always correct): `_version` in
                # _frames_key catches in-place refills through torch ops, but not writes through `.data`, DLPack / numpy views or a
                # native kernel's pointer, and a stale pack would train on the previous minibatch without any error.
                nb_, na_ = next_frames[0], next_frames[1]
                aliased = nb_.data_ptr() == before.data_ptr() or (na_ is not None and after is not None and na_.data_ptr() == after.data_ptr())
                if not aliased or (len(next_frames) > 3 and next_frames[3] is True):
                    self._pack_ahead(next_frames[:3], 1 if slot == 0 else 0)
            if not n.extra_capacity:  # train-mode BatchNorm: model(before) [+ model(after)], F feature calls each
                n.num_batches_tracked += (1 if self.gtb else 2) * n.num_frames
                n.mark_dirty()  # the running statistics changed: eval-mode packed weights are stale
            for stage in range(3):
                _lib.check(self.lib.vdqn_net_backward_stage(n.handle, C.byref(a), stage, st), "vdqn_net_backward_stage")
                if self.allreduce is not None:
                    # the stage's gradients are complete on the engine's gradient stream, not on `st` (which is already
                    # running the next stage's data gradients): the collective is ordered behind that stream
                    b, e = self.stage_ranges[stage]
                    with self._grad_stream_ctx():
                        self.allreduce(self.grads[b:e], stage)
                        if stage == 2 and self.allreduce_loss is not None:
                            self.allreduce_loss(self.loss)
                    if _DIST_EARLY_ADAM and stage < 2 and self.allreduce_wait is not None and n.extra_capacity and self._dist_early_ok():
                        b4, e4 = (b + 3) // 4 * 4, e // 4 * 4
                        if e4 > b4:
                            if self._post_stream is None:
                                self._post_stream = torch.cuda.Stream(device=n.device)
                            with torch.cuda.stream(self._post_stream):
                                if _DIST_EARLY_ADAM_MODE != "2":
                                    self.allreduce_wait()  # this bucket's collective, nothing else
                                else:  # (diagnostic: ordered behind the gradient stream only)
                                    torch.cuda.current_stream().wait_stream(self._grad_stream)
                                if _DIST_EARLY_ADAM_MODE != "3":
                                    self._adam_range(b4, e4, self.adam_step + 1)
                            if _DIST_EARLY_ADAM_MODE != "3":
                                self._adam_done.append((b4, e4))
                            self._post_used = True
                elif early_adam and stage < 2:
                    b, e = self.stage_ranges[stage]
                    b, e = (b + 3) // 4 * 4, e // 4 * 4  # vdqn_adam wants 16-byte aligned ranges; the rest is left to optimizer_step
                    if e > b:
                        with self._grad_stream_ctx():
                            self._adam_range(b, e, self.adam_step + 1)
                        self._adam_done.append((b, e))
        del keep

    def _grad_stream_ctx(self):
        ptr = self.lib.vdqn_net_grad_stream(self.net.handle)
        if not ptr:
            return contextlib.nullcontext()
        if self._grad_stream is None or self._grad_stream.cuda_stream != ptr:
            self._grad_stream = torch.cuda.ExternalStream(ptr, device=self.net.device)
        return torch.cuda.stream(self._grad_stream)

    def _adam_range(self, b: int, e: int, step: int):
        """Adam (train_q_network.py:227) over the flat element range [b, e) on the current stream."""
        n = self.net
        _lib.check(self.lib.vdqn_adam(n.params.data_ptr() + 4 * b, self.grads.data_ptr() + 4 * b, self.exp_avg.data_ptr() + 4 * b,
                                      self.exp_avg_sq.data_ptr() + 4 * b, e - b, step, self.lr, self.betas[0], self.betas[1],
                                      self.eps, _stream()), "vdqn_adam")

    def optimizer_step(self):
        n = self.net
        self.adam_step += 1
        done = sorted(getattr(self, "_adam_done", []))
        self._adam_done = []
        with torch.cuda.device(n.device):
            if self._post_used:  # per-bucket updates under an exchange ran on their own stream: join it
                torch.cuda.current_stream().wait_stream(self._post_stream)
                self._post_used = False
            pos = 0  # everything of [0, trainable_numel) that `forward_backward(early_adam=True)` has not updated already
            for b, e in done + [(n.trainable_numel, n.trainable_numel)]:
                if b > pos:
                    self._adam_range(pos, b, self.adam_step)
                pos = max(pos, e)
        n.mark_dirty()

    def step(self, before, after, src_kind, act, rew, term, valid=None, gt=None, finish_allreduce=None, next_frames=None) -> torch.Tensor:
        """One iteration of the reference loop body (train_q_network.py:213-227).  Returns the device loss scalar
        (no host sync).

        next_frames = (before, after, src_kind) of the NEXT call, if the loop already has them (the reference's DataLoader does: it
        prefetches): they are normalised and packed for the stem while this update's head and backward pass run, instead of at
        the start of the next update.  The tensors must not change until that call (refills must go through version-bumping torch
        ops such as copy_; tensors that alias THIS call's are accepted only as (before, after, src_kind, True) — the caller's promise
        that they are replayed unchanged); a call whose frames are not the ones announced packs its own, as always.  Same arithmetic, same results — and, measured, a SLOWER update (5.81 against 5.74 ms,
        profiles/r03w_ab_pack_ahead.txt: the HBM-bound pack beside layer4 and the head costs more than the start of the update
        gains), so neither bench.py nor the trainer uses it by default; it stays for callers whose frames arrive packed
        (vdqn_step_args.packed_frames)."""
        self.sample_number += 1
        if self.sample_number % self.tui == 0:
            self.sync_target()
        # (single process only: behind each RCCL bucket on a stream of its own it measured 7.22 vs 5.96 ms per update with one rank,
        # profiles/r03s_ab_rccl_early_adam.txt — with an exchange the whole optimiser update stays behind `finish_allreduce`)
        early = _EARLY_ADAM and self.net.extra_capacity and self.allreduce is None and finish_allreduce is None
        self.forward_backward(before, after, src_kind, act, rew, term, valid, gt, early_adam=early, next_frames=next_frames)
        if finish_allreduce is not None:
            finish_allreduce()
        self.optimizer_step()
        return self.loss
