"""Rank program for tests/test_launch_cpu.py: joins a gloo group from the environment the launcher set, sums the ranks,
rank 0 prints ONE JSON line.  `--fail-rank R` makes rank R exit 3 before the group forms; `--hang-rank R` makes rank R write
its pid to `--pid-dir`/rank<R>.pid and sleep for ever (a peer stuck in a collective) while the others skip the group and
exit 0 (`--others-exit`) or sleep too."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from video_dqn_amd import launch  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[sys.argv.index("--gpus") + 1])
    launch.die_with_parent()  # a rank started by spawn_ranks goes down with its launcher
    if n > 1 and not launch.in_rank_env():
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], n))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(3)
    if "--hang-rank" in sys.argv:
        import time
        pid_dir = sys.argv[sys.argv.index("--pid-dir") + 1]
        with open(os.path.join(pid_dir, f"rank{rank}.pid"), "w") as f:
            f.write(str(os.getpid()))
        if rank == int(sys.argv[sys.argv.index("--hang-rank") + 1]) or "--others-exit" not in sys.argv:
            while True:
                time.sleep(1)
        sys.exit(0)
    out_stream = launch.claim_stdout()
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank != 0:
        print("noise from a non-zero rank")  # must not reach the parent's stdout
    else:
        print(json.dumps({"n_gpus": world, "sum": t.item(), "local_rank": int(os.environ["LOCAL_RANK"])}), file=out_stream, flush=True)
    dist.barrier()
    dist.destroy_process_group()
