"""The self-launcher (video_dqn_amd/launch.py) that bench.py --gpus N and train_q_network.py -g a,b,.. use when no
torchrun is around them: N rank processes from a parent that never touches the GPU, rank 0's stdout relayed, a failing
rank ends the job with its exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "scripts", "rank_echo.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def test_spawn_ranks_relays_rank0_only():
    r = subprocess.run([sys.executable, SCRIPT, "--gpus", "2"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE line: rank 1's print went to stderr
    out = json.loads(lines[0])
    assert out == {"n_gpus": 2, "sum": 3.0, "local_rank": 0}
    assert "noise from a non-zero rank" in r.stderr


def test_failed_rank_fails_the_job():
    r = subprocess.run([sys.executable, SCRIPT, "--gpus", "2", "--fail-rank", "1"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=240)
    assert r.returncode == 3
    assert r.stdout.strip() == ""


def test_parent_of_bench_and_cli_never_imports_the_gpu_runtime_before_spawning():
    """The spawn decision sits above every GPU call in both entry points (static check: the launcher call precedes the
    first torch.cuda / _lib use in main)."""
    for name, first_gpu_use in (("bench.py", "torch.cuda.is_available()"), ("train_q_network.py", "import torch")):
        src = open(os.path.join(ROOT, name)).read()
        main = src[src.index("def main()"):] if "def main()" in src else src[src.index('if __name__ == "__main__"'):]
        assert "launch.spawn_ranks(" in main
        assert main.index("launch.spawn_ranks(") < main.index(first_gpu_use), name
    assert "import torch" not in open(os.path.join(ROOT, "video_dqn_amd", "launch.py")).read().replace("imports torch", "")
