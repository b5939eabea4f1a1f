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


def test_ppo_loop_line_with_two_ranks_and_stub_task():
    """`bench.py --gpus 2 --ppo` (BASELINE config 5's launch shape): rank spawn, per-epoch time aggregation (MAX over ranks) and the line."""
    d, err = _run(["--gpus", "2", "--ppo", "--ppo-epochs", "2", "--stub-task", "--num-envs", "128"])
    assert d["n_gpus"] == 2 and d["config"]["world_size_seen"] == 2 and d["config"]["world_size_matches_gpus"] and d["config"]["backend"] == "gloo"
    assert d["steps"] == 64 and d["config"]["global_envs"] == 256 and d["config"]["fps_total"] <= d["config"]["fps_step"]
    assert "world 2" in err and "no multi-GPU curve" in d["config"]["scaling_curve"]


def test_self_launch_eight_ranks_with_stub_task():
    """`bench.py --gpus 8` - the driver's widest launch - through the rank spawn, the barrier, the MAX over ranks and the one line (gloo,
    stub task: the launch logic, never a measurement).  No 8-GPU node has been available to any round; this is what pins the N = 8 path."""
    d, _ = _run(["--gpus", "8", "--stub-task", "--steps", "64", "--warmup", "32", "--num-envs", "64"])
    assert d["n_gpus"] == 8 and d["config"]["world_size_seen"] == 8 and d["config"]["world_size_matches_gpus"] and d["config"]["backend"] == "gloo"
    assert len(d["config"]["per_rank_env_steps_per_s"]) == 8 and len(d["config"]["per_rank_kernel_ms"]) == 8 and d["config"]["global_envs"] == 512
    assert d["value"] <= sum(d["config"]["per_rank_env_steps_per_s"]) * (1 + 1e-9) and d["value"] >= 8 * min(d["config"]["per_rank_env_steps_per_s"]) * (1 - 1e-9)
    assert d["metric"].startswith("env-steps/sec at num_envs=64") and d["scaling"] == "weak"


def test_ppo_loop_line_with_eight_ranks_and_stub_task():
    """BASELINE config 5's launch shape: `bench.py --gpus 8 --ppo` (stub task and stub agent over gloo)."""
    d, err = _run(["--gpus", "8", "--ppo", "--ppo-epochs", "2", "--stub-task", "--num-envs", "64"])
    assert d["n_gpus"] == 8 and d["config"]["world_size_seen"] == 8 and d["config"]["world_size_matches_gpus"] and d["config"]["global_envs"] == 512
    assert d["config"]["fps_total"] <= d["config"]["fps_step"] and "world 8" in err


def test_short_run_is_followed_by_a_whole_epoch_block():
    """The driver's command (--steps 20 --warmup 5): every run() starts an epoch, so the timed region is reset + positions 0..19 - the
    line must say so, and carry a separately timed block of whole epochs from which the roofline figures are taken."""
    d, err = _run(["--stub-task", "--steps", "20", "--warmup", "5", "--num-envs", "64"])
    assert d["n_gpus"] == 1 and d["config"]["world_size_seen"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["timed_epoch_positions"] == "0..19" and d["config"]["resets_in_timed_region"] == 1
    assert d["config"]["timed_steps_cover_whole_epochs"] is False
    w = d["whole_epoch"]
    assert w["steps"] == 320 and w["epochs"] == 10 and w["kernel_launches_timed"] == 320
    assert d["roofline"]["kernel_ms"] == w["kernel_ms"] and d["roofline"]["requested_region"]["kernel_launches_timed"] == 20
    # the headline is the rollout average (the whole-epoch block); the requested 20 steps - the light start of an epoch - stay beside it
    assert d["value"] == w["value"] and d["ms_per_step"] == w["ms_per_step"] and "whole_epoch block" in d["value_region"]
    r = d["requested_region"]
    assert r["steps"] == 20 and r["epoch_positions"] == "0..19" and abs(r["ms_per_step"] * 20 * 1e-3 - 20 * 64 / r["value"]) < 1e-6
    assert "not a whole number" in err


def test_whole_epoch_runs_need_no_extra_block():
    d, _ = _run(["--stub-task", "--steps", "64", "--warmup", "32", "--num-envs", "64"])
    assert "whole_epoch" not in d and d["config"]["timed_epoch_positions"] == "all, 2 times" and d["config"]["resets_in_timed_region"] == 2


def test_gpu_run_refuses_without_gpus():
    import torch

    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU" in (p.stderr + p.stdout)
