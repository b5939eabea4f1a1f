"""bench.py --gpus N launches its own ranks (CPU test of the launch / collection logic with a stub task over gloo; the measured
path needs GPUs and is covered by the driver's runs)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_command_is_the_drivers_launch_line():
    sys.path.insert(0, REPO)
    import bench

    cmd = bench.rank_command(8, ["--gpus", "8", "--steps", "20"], port=1234)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[4:6] == ["--nproc-per-node", "8"] and cmd[6:10] == ["--master-addr", "127.0.0.1", "--master-port", "1234"]
    assert cmd[10].endswith("bench.py") and cmd[11:] == ["--gpus", "8", "--steps", "20"]


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_self_launch_two_ranks_with_stub_task():
    d, _ = _run(["--gpus", "2", "--stub-task", "--steps", "64", "--warmup", "32", "--num-envs", "256"])
    assert d["n_gpus"] == 2 and d["config"]["world_size_seen"] == 2 and d["config"]["backend"] == "gloo"
    assert len(d["config"]["per_rank_env_steps_per_s"]) == 2
    assert d["steps"] == 64 and d["warmup"] == 32 and d["scaling"] == "weak" and d["config"]["global_envs"] == 512
    # whole-job value = all envs of all ranks over the slowest rank's time
    assert d["value"] <= sum(d["config"]["per_rank_env_steps_per_s"]) * (1 + 1e-9)
    assert d["value"] >= 2 * min(d["config"]["per_rank_env_steps_per_s"]) * (1 - 1e-9)
    assert d["roofline"]["kernel_launches_timed"] == 64 and d["ms_per_step"] >= d["roofline"]["kernel_ms"]
    assert "STUB" in d["config"]["workload"] and "cpu_baseline" not in d


def test_single_rank_and_partial_epoch_warning():
    d, err = _run(["--stub-task", "--steps", "20", "--warmup", "5", "--num-envs", "64"])
    assert d["n_gpus"] == 1 and d["config"]["world_size_seen"] == 1
    assert "not a multiple" in err


def test_gpu_run_refuses_without_gpus():
    import torch

    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU" in (p.stderr + p.stdout)
