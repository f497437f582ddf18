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


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # a zombie still answers kill(0): read its state
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except FileNotFoundError:
        return False


def _wait_pids(pid_dir, n, timeout=60):
    import time
    t0 = time.monotonic()
    while time.monotonic() - t0 < timeout:
        names = [f for f in os.listdir(pid_dir) if f.endswith(".pid")]
        if len(names) >= n:
            pids = []
            for f in names:
                txt = open(os.path.join(pid_dir, f)).read().strip()
                if txt:
                    pids.append(int(txt))
            if len(pids) >= n:
                return pids
        time.sleep(0.1)
    raise AssertionError("ranks did not start")


def test_ranks_die_with_a_signalled_launcher(tmp_path):
    """SIGTERM to the launcher (gpurun --timeout, a pytest timeout): every rank is stopped, the exit code is 128 + 15."""
    import signal
    import time
    p = subprocess.Popen([sys.executable, SCRIPT, "--gpus", "2", "--hang-rank", "0", "--pid-dir", str(tmp_path)], env=_clean_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    pids = _wait_pids(str(tmp_path), 2)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    time.sleep(0.2)
    assert not any(_alive(q) for q in pids)


def test_ranks_die_with_a_killed_launcher(tmp_path):
    """SIGKILL to the launcher: no handler runs; the ranks get SIGTERM from the kernel (PR_SET_PDEATHSIG)."""
    import signal
    import time
    p = subprocess.Popen([sys.executable, SCRIPT, "--gpus", "2", "--hang-rank", "0", "--pid-dir", str(tmp_path)], env=_clean_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    pids = _wait_pids(str(tmp_path), 2)
    p.kill()
    p.wait(timeout=60)
    t0 = time.monotonic()
    while any(_alive(q) for q in pids) and time.monotonic() - t0 < 30:
        time.sleep(0.1)
    assert not any(_alive(q) for q in pids)


def test_straggler_behind_a_finished_rank_is_bounded(tmp_path):
    """One rank exits 0 while its peer hangs (stuck in a collective): the job ends with 124 after the grace period."""
    env = _clean_env()
    env["VDQN_LAUNCH_STRAGGLER_GRACE"] = "1.5"
    r = subprocess.run([sys.executable, SCRIPT, "--gpus", "2", "--hang-rank", "1", "--others-exit", "--pid-dir", str(tmp_path)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 124
    assert not any(_alive(q) for q in _wait_pids(str(tmp_path), 2))


def test_launch_timeout(tmp_path):
    env = _clean_env()
    env["VDQN_LAUNCH_TIMEOUT"] = "2"
    r = subprocess.run([sys.executable, SCRIPT, "--gpus", "2", "--hang-rank", "0", "--pid-dir", str(tmp_path)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 124
    assert not any(_alive(q) for q in _wait_pids(str(tmp_path), 2))


def test_parent_of_bench_and_cli_never_imports_the_gpu_runtime_before_spawning():
    """The spawn decision sits above every GPU call in both entry points (static check: the launcher call precedes the
    first torch.cuda / _lib use in main)."""
    for name, first_gpu_use in (("bench.py", "torch.cuda.is_available()"), ("train_q_network.py", "import torch")):
        src = open(os.path.join(ROOT, name)).read()
        main = src[src.index("def main()"):] if "def main()" in src else src[src.index('if __name__ == "__main__"'):]
        assert "launch.spawn_ranks(" in main
        assert main.index("launch.spawn_ranks(") < main.index(first_gpu_use), name
    assert "import torch" not in open(os.path.join(ROOT, "video_dqn_amd", "launch.py")).read().replace("imports torch", "")
