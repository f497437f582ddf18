#!/usr/bin/env python
"""Is torch.distributed's Work.wait() (backend nccl = RCCL) host-blocking on this build?  One rank; an all-reduce is launched
(async_op) behind a long-running kernel on a side stream, and wait() is called from another stream right away: a device-side wait
returns in microseconds, a host-blocking one after the long kernel and the collective have finished.  (Diagnosis of the +1.3 ms
that per-bucket Adam under an exchange costs with one rank: profiles/r03s_ab_rccl_early_adam.txt, r05g_abd_*.)"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.zeros(10 * 1024 * 1024, device=dev)
big = torch.randn(8192, 8192, device=dev)
side, post = torch.cuda.Stream(), torch.cuda.Stream()
dist.all_reduce(x)  # communicator set-up
torch.cuda.synchronize()
for trial in range(3):
    with torch.cuda.stream(side):
        t0 = time.perf_counter()
        for _ in range(20):
            y = big @ big  # ~20 x 0.6 ms of work in front of the collective
        t_q = time.perf_counter()
        w = dist.all_reduce(x, async_op=True)
        t_l = time.perf_counter()
    with torch.cuda.stream(post):
        w.wait()
        t_w = time.perf_counter()
        x.add_(1.0)
    t_a = time.perf_counter()
    torch.cuda.synchronize()
    t_s = time.perf_counter()
    print(f"trial {trial}: queue matmuls {1e3 * (t_q - t0):.2f} ms, launch all_reduce {1e3 * (t_l - t_q):.2f} ms, Work.wait() {1e3 * (t_w - t_l):.3f} ms on the host, "
          f"queue add {1e3 * (t_a - t_w):.3f} ms, then device sync {1e3 * (t_s - t_a):.2f} ms")
dist.destroy_process_group()
