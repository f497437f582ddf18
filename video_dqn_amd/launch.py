"""One process per GPU, started by the program itself: ``spawn_ranks`` re-runs the calling script N times with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the environment ``torch.distributed.run`` would give
them) and relays rank 0's stdout.  The parent never touches the GPU — nothing here imports torch — so no process
that has initialised HIP ever execs or forks.  A failed rank ends the others and its exit code is returned; there
is no in-process retry.

The reference has no launcher (single GPU: train_q_network.py:255-259,275); this is the entry the N > 1 path needs.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def in_rank_env() -> bool:
    """True inside a rank process (ours or torch.distributed.run's)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(argv: Sequence[str], world_size: int, extra_env: Optional[Dict[str, str]] = None,
                timeout_s: Optional[float] = None) -> int:
    """Run ``python argv...`` as ``world_size`` rank processes; returns 0 or the first non-zero exit code.

    Rank 0 inherits stdout (its ONE JSON line / progress output is the program's output); the other ranks' stdout is
    sent to stderr so nothing they print can be mistaken for the result."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs: List[subprocess.Popen] = []
    for rank in range(world_size):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world_size), "LOCAL_WORLD_SIZE": str(world_size),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=None if rank == 0 else sys.stderr))
    t0 = time.monotonic()
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
        if rc != 0 or (timeout_s is not None and time.monotonic() - t0 > timeout_s):
            if rc == 0:
                rc = 124
            for p in live:  # exactly the processes started above, by handle
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if live:
            time.sleep(0.05)
    return rc


def claim_stdout():
    """Keep the process's real stdout for the ONE result line: returns a text stream on a duplicate of fd 1 and points fd 1
    at stderr, so anything a native library prints to stdout (gloo's "[Gloo] Rank 0 is connected ..." banner) cannot
    land next to the JSON line."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real
