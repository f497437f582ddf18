"""One process per GPU, started by the program itself: ``spawn_ranks`` re-runs the calling script N times with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the environment ``torch.distributed.run`` would give
them) and relays rank 0's stdout.  The parent never touches the GPU — nothing here imports torch — so no process
that has initialised HIP ever execs or forks.  A failed rank ends the others and its exit code is returned; there
is no in-process retry.  The ranks never outlive the launcher (signals forwarded, try/finally, PR_SET_PDEATHSIG, a
bounded wait for stragglers once one rank has finished): see ``spawn_ranks``.

The reference has no launcher (single GPU: train_q_network.py:255-259,275); this is the entry the N > 1 path needs.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def in_rank_env() -> bool:
    """True inside a rank process (ours or torch.distributed.run's)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def die_with_parent() -> None:
    """Called by a rank program at start-up (bench.py, train_q_network.py, before they touch the GPU): if this process was
    started by ``spawn_ranks`` it gets SIGTERM when the launcher dies, however that happens.  Done here rather than in a
    ``preexec_fn`` of the launcher, which is not safe once the launching process has threads (it has: torch is imported)."""
    ppid = os.environ.get("VDQN_LAUNCHER_PID")
    if not ppid or not in_rank_env():
        return
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)  # PR_SET_PDEATHSIG
        if os.getppid() != int(ppid):  # the launcher died before the prctl took effect
            os.kill(os.getpid(), signal.SIGTERM)
    except Exception:
        pass


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _stop(procs: Sequence[subprocess.Popen], grace_s: float = 20.0) -> None:
    """Terminate exactly the given children (by handle; each is the leader of its own session, so its process group is the
    rank and whatever the rank started), wait, then kill what is left."""
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
    t_end = time.monotonic() + grace_s
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def spawn_ranks(argv: Sequence[str], world_size: int, extra_env: Optional[Dict[str, str]] = None,
                timeout_s: Optional[float] = None, straggler_grace_s: float = 300.0) -> int:
    """Run ``python argv...`` as ``world_size`` rank processes; returns 0 or the first non-zero exit code.

    Rank 0 inherits stdout (its ONE JSON line / progress output is the program's output); the other ranks' stdout is
    sent to stderr so nothing they print can be mistaken for the result.

    No rank outlives this call: the ranks are stopped (SIGTERM to each rank's own session, SIGKILL after 20 s) when one of
    them fails, when ``timeout_s`` (default: ``VDQN_LAUNCH_TIMEOUT`` seconds, unset = none) runs out, when a rank is still
    running ``straggler_grace_s`` after the first one exited cleanly (a peer stuck in a collective), when this process gets
    SIGTERM / SIGINT / SIGHUP (returns 128 + signal), and on any exception (try/finally).  Should this process die without
    running any of that (SIGKILL), the ranks still go: each rank program calls ``die_with_parent()`` first thing, which asks
    the kernel for SIGTERM on the launcher's death (PR_SET_PDEATHSIG; VDQN_LAUNCHER_PID tells it who the launcher is)."""
    if timeout_s is None and os.environ.get("VDQN_LAUNCH_TIMEOUT"):
        timeout_s = float(os.environ["VDQN_LAUNCH_TIMEOUT"])
    if os.environ.get("VDQN_LAUNCH_STRAGGLER_GRACE"):
        straggler_grace_s = float(os.environ["VDQN_LAUNCH_STRAGGLER_GRACE"])
    port = os.environ.get("MASTER_PORT") or str(free_port())
    parent = os.getpid()

    procs: List[subprocess.Popen] = []
    got_signal: List[int] = []
    old_handlers = {}

    def _on_signal(signum, _frame):
        got_signal.append(signum)

    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            old_handlers[sg] = signal.signal(sg, _on_signal)
        except ValueError:  # not the main thread: the finally clause below still stops the ranks
            pass
    rc = 0
    try:
        for rank in range(world_size):
            env = dict(os.environ)
            # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool only support dmabuf IPC; with the legacy mode RCCL's peer
            # buffer exchange (and any HIP tensor shared across processes) fails with "hipIpcGetMemHandle: invalid argument".
            # The image exports it already; it is repeated here so that a rank started from a scrubbed environment (pytest's
            # subprocess env, a service manager) still gets it.  A value the caller set wins.
            env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world_size), "LOCAL_WORLD_SIZE": str(world_size),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                        "VDQN_LAUNCHER_PID": str(parent)})
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=None if rank == 0 else sys.stderr,
                                          start_new_session=True))  # (no preexec_fn: unsafe once the parent has threads)
        t0 = time.monotonic()
        first_clean_exit: Optional[float] = None
        live = list(procs)
        while live:
            for p in list(live):
                r = p.poll()
                if r is None:
                    continue
                live.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                if r == 0 and first_clean_exit is None:
                    first_clean_exit = time.monotonic()
            now = time.monotonic()
            if got_signal:
                rc = 128 + got_signal[0]
            elif rc == 0 and live and timeout_s is not None and now - t0 > timeout_s:
                rc = 124
            elif rc == 0 and live and first_clean_exit is not None and now - first_clean_exit > straggler_grace_s:
                rc = 124
            if rc != 0:
                break
            if live:
                time.sleep(0.05)
    finally:
        _stop(procs)
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    return rc


def claim_stdout():
    """Keep the process's real stdout for the ONE result line: returns a text stream on a duplicate of fd 1 and points fd 1
    at stderr, so anything a native library prints to stdout (gloo's "[Gloo] Rank 0 is connected ..." banner) cannot
    land next to the JSON line."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real
