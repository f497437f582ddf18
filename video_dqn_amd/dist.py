"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).  The reference has no distributed code (single GPU, train_q_network.py:255-259,275);
this is the one exchange step the path needs: a SUM all-reduce of the flat f32 gradient.  Because the fused
TD kernel already divides by the *global* batch (inv_count = 1 / (5 * B * world)), the sum is the exact
gradient of the global-batch mean loss (train_q_network.py:180) — no extra scaling pass.

The flat gradient is laid out by backward stage (head+layer4 | layer3 | layer2+layer1+stem), so each bucket is
one contiguous slice that is complete as soon as its stage's unfold kernel has run; ``launch`` is called right
after each stage and issues an asynchronous all-reduce that overlaps the next stage's kernels, ``finish``
makes the compute stream wait for all buckets before Adam.  xGMI is point-to-point, 49.7 MB total: three large
buckets rather than per-layer messages.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class BucketAllReduce:
    def __init__(self, world_size: int, group=None, force: bool = False):
        self.world_size = world_size
        self.group = group
        self.force = force  # issue the collectives even at world size 1 (exercises RCCL + the stream ordering on one GPU)
        self._works: List = []
        self.bucket_bytes: List[int] = []  # sizes of the buckets reduced in the last update, in launch order
        self._loss_bufs = [None, None]     # launch_loss: two device scalars used in turn
        self._loss_slot = 0
        self._loss_pending = None
        self._loss_copied = None           # event behind launch_loss's read of the stepper's loss scalar (finish() waits for it)

    def launch_loss(self, loss: torch.Tensor) -> None:
        """SUM the ranks' shares of the global-batch mean loss (train_q_network.py:180,228-231) WITHOUT the compute stream ever
        waiting for it: called behind the last gradient bucket (the stream on which that bucket was launched is current), the
        4-byte collective goes into a buffer of its own and only `take_loss`'s consumer — the trainer's loss read-back stream,
        one update late — waits for it."""
        if self.world_size == 1 and not self.force:
            return
        self._loss_slot ^= 1
        buf = self._loss_bufs[self._loss_slot]
        if buf is None or buf.device != loss.device:
            buf = self._loss_bufs[self._loss_slot] = torch.zeros(1, dtype=torch.float32, device=loss.device)
        buf.copy_(loss.reshape(1))
        # the NEXT update zeroes `loss` on the compute stream: finish() orders that stream behind this 4-byte read (not behind the
        # collective), so the read / overwrite pair is ordered whatever the schedule of the target pass is
        if loss.is_cuda:
            self._loss_copied = torch.cuda.Event()
            self._loss_copied.record()
        self._loss_pending = (buf, dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def take_loss(self):
        """-> (device scalar, async work) of the last `launch_loss`, or None.  work.wait() orders the CURRENT stream behind the
        collective: call it on the stream that reads the value, not on the compute stream."""
        p, self._loss_pending = self._loss_pending, None
        return p

    def launch(self, grad_slice: torch.Tensor, stage: int) -> None:
        if self.world_size == 1 and not self.force:
            return
        if stage == 0:
            self.bucket_bytes = []
        self.bucket_bytes.append(grad_slice.numel() * grad_slice.element_size())
        self._works.append(dist.all_reduce(grad_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait_last(self) -> None:
        """Order the CURRENT stream behind the most recently launched bucket (a consumer of that bucket alone — the per-bucket
        optimiser update of TDStepper(dist_early_adam=True) — on a stream of its own; `finish` still joins everything)."""
        if self._works:
            self._works[-1].wait()

    def finish(self) -> None:
        for w in self._works:
            w.wait()
        self._works.clear()
        if self._loss_copied is not None:
            torch.cuda.current_stream().wait_event(self._loss_copied)
            self._loss_copied = None


class CAbiBucketAllReduce:
    """The same exchange through the C ABI's own RCCL entries (include/vdqn.h: vdqn_comm_init / vdqn_allreduce_bucket) instead
    of torch.distributed — what a caller that binds the header without PyTorch's process group does.  The 128-byte
    rendezvous id travels through a file (`uid_path` + a per-job nonce — MASTER_PORT / VDQN_JOB_NONCE, so an id left behind by an
    earlier or crashed job at the same path is never picked up; rank 0 writes it, the others wait for it, rank 0 removes it once
    every rank has joined).  Same contract as
    BucketAllReduce: ``launch`` is called with the stream current on which the stage's gradients are complete and queues the
    collective there; ``finish`` makes the then-current stream wait for all queued collectives."""

    def __init__(self, rank: int, world_size: int, uid_path: str, force: bool = False, timeout_s: float = 120.0):
        import ctypes as C
        import os
        import time
        from . import _lib
        self._lib, self._C = _lib, C
        self.lib = _lib.load()
        self.world_size, self.force = world_size, force
        self.bucket_bytes: List[int] = []
        self._events: List = []
        self._loss_bufs, self._loss_slot, self._loss_pending, self._loss_copied = [None, None], 0, None, None
        uid = C.create_string_buffer(_lib.COMM_UID_BYTES)
        nonce = os.environ.get("VDQN_JOB_NONCE") or os.environ.get("MASTER_PORT") or os.environ.get("VDQN_LAUNCHER_PID") or "0"
        uid_path = f"{uid_path}.{nonce}"
        self._uid_path = uid_path if rank == 0 else None
        if rank == 0:
            _lib.check(self.lib.vdqn_comm_unique_id(uid), "vdqn_comm_unique_id")
            tmp = uid_path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(uid.raw)
            os.replace(tmp, uid_path)
        else:
            t0 = time.monotonic()
            while not os.path.exists(uid_path):
                if time.monotonic() - t0 > timeout_s:
                    raise _lib.VdqnError(f"rank {rank}: no rendezvous id at {uid_path} after {timeout_s} s")
                time.sleep(0.02)
            uid.raw = open(uid_path, "rb").read()
        self.handle = C.c_void_p()
        _lib.check(self.lib.vdqn_comm_init(rank, world_size, uid, C.byref(self.handle)), "vdqn_comm_init")
        if self._uid_path:  # ncclCommInitRank returns once every rank has joined: the id has served its purpose
            try:
                os.unlink(self._uid_path)
            except OSError:
                pass

    def launch(self, grad_slice: torch.Tensor, stage: int) -> None:
        if self.world_size == 1 and not self.force:
            return
        if stage == 0:
            self.bucket_bytes = []
        self.bucket_bytes.append(grad_slice.numel() * grad_slice.element_size())
        st = torch.cuda.current_stream()
        self._lib.check(self.lib.vdqn_allreduce_bucket(self.handle, grad_slice.data_ptr(), grad_slice.numel(), self._lib.VDQN_F32,
                                                       st.cuda_stream), "vdqn_allreduce_bucket")
        ev = torch.cuda.Event()
        ev.record(st)
        self._events.append(ev)

    def wait_last(self) -> None:
        if self._events:
            torch.cuda.current_stream().wait_event(self._events[-1])

    def finish(self) -> None:
        cur = torch.cuda.current_stream()
        for ev in self._events:
            cur.wait_event(ev)
        self._events.clear()
        if self._loss_copied is not None:
            cur.wait_event(self._loss_copied)
            self._loss_copied = None

    class _EventWork:  # the `work` of take_loss(): wait() orders the current stream behind the collective
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def launch_loss(self, loss: torch.Tensor) -> None:
        """As BucketAllReduce.launch_loss: the loss shares are summed on the gradient stream behind the last bucket."""
        if self.world_size == 1 and not self.force:
            return
        self._loss_slot ^= 1
        buf = self._loss_bufs[self._loss_slot]
        if buf is None:
            buf = self._loss_bufs[self._loss_slot] = torch.zeros(1, dtype=torch.float32, device=loss.device)
        buf.copy_(loss.reshape(1))
        st = torch.cuda.current_stream()
        self._loss_copied = torch.cuda.Event()  # (see BucketAllReduce.launch_loss)
        self._loss_copied.record(st)
        self._lib.check(self.lib.vdqn_allreduce_bucket(self.handle, buf.data_ptr(), 1, self._lib.VDQN_F32, st.cuda_stream), "vdqn_allreduce_bucket")
        ev = torch.cuda.Event()
        ev.record(st)
        self._loss_pending = (buf, self._EventWork(ev))

    def take_loss(self):
        p, self._loss_pending = self._loss_pending, None
        return p

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.vdqn_comm_destroy(self.handle)
            self.handle = None


def shard_indices(n_items: int, rank: int, world_size: int, batch_per_rank: int, epoch_perm) -> list:
    """Rank-strided, drop_last sharding of one shuffled epoch (mirrors DataLoader(shuffle=True, drop_last=True),
    train_q_network.py:98,114): rank r takes perm[r::world]; every rank gets the same number of full batches."""
    mine = epoch_perm[rank::world_size]
    per_rank = (n_items // world_size // batch_per_rank) * batch_per_rank
    return list(mine[:per_rank])


def broadcast_replica_state(tensors, src: int = 0, group=None) -> None:
    """Make every rank's copy of the given tensors (master parameters, BatchNorm statistics, Adam moments) equal to
    rank `src`'s.  Seeding every rank alike already gives equal initial weights; the broadcast makes that a guarantee
    (and covers a resume where only rank 0 found the checkpoint readable)."""
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def agree_all(flag: bool, device=None, group=None) -> bool:
    """True only if `flag` is true on EVERY rank (MIN all-reduce): per-rank decisions that must match across the job —
    e.g. whether the decoded frames fit in this rank's HBM — are taken through this, so no two ranks pick different
    data paths."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())
