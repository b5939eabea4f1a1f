"""Two ranks of the REAL rollout engine (one process each, both on cuda:0, gloo for the collectives - RCCL needs one GPU per rank):
env sharding with per-rank seeds (run.py:37), no collective in the rollout, the PPO-side exchange on engine data."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ENVS_PER_RANK = 96
STEPS = 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from vid2player3d_amd.dist import all_gather_advantages, global_advantage_stats

    task = bench.build_task(ENVS_PER_RANK, 0, seed=7 + rank, substep_jobs=True)  # per-rank seed like run.py:37
    dev = task.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)
    task.reset()
    rews, alive = [], []
    for k in range(STEPS):
        task.step_fused(bench.make_actions(task, 0.17 * torch.randn((ENVS_PER_RANK, 75), device=dev, generator=gen)))
        rews.append(task.rew_buf.clone())
        alive.append((task.reset_buf == 0).float())
    task.check()
    rew, msk = torch.stack(rews).cpu(), torch.stack(alive).cpu()  # [T, N_local] (gloo: CPU tensors)
    gathered = all_gather_advantages(rew)
    mean, std, cnt = global_advantage_stats(rew, msk)
    q.put((rank, rew.numpy(), msk.numpy(), gathered.numpy(), float(mean), float(std), float(cnt), task.obs_buf.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()
    task.close()


def test_two_ranks_of_the_engine():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, rew0, m0, g0, mean0, std0, cnt0, obs0), (_, rew1, m1, g1, mean1, std1, cnt1, obs1) = res
    assert np.isfinite(obs0).all() and np.isfinite(obs1).all()
    assert rew0.shape == (STEPS, ENVS_PER_RANK) and (rew0 > 0).any() and (rew1 > 0).any()
    assert not np.allclose(rew0, rew1), "the ranks roll out different envs (seed 7 + rank)"
    full = np.concatenate([rew0, rew1], axis=1)
    assert np.array_equal(g0, full) and np.array_equal(g1, full), "all-gather: envs concatenated in rank order on every rank"
    msk = np.concatenate([m0, m1], axis=1).astype(np.float64)
    mean = (full * msk).sum() / msk.sum()
    std = np.sqrt(((full - mean) ** 2 * msk).sum() / (msk.sum() - 1))
    for m, s, c in ((mean0, std0, cnt0), (mean1, std1, cnt1)):
        assert abs(m - mean) < 1e-5 and abs(s - std) < 1e-5 and c == msk.sum()


def _rank_ppo(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from vid2player3d_amd.ppo import PPOAgent

    task = bench.build_task(ENVS_PER_RANK, 0, seed=7 + rank, substep_jobs=True)
    agent = PPOAgent(task, units=(64, 32), minibatch_envs=32, mini_epochs=2, seed=1, learning_rate=1e-3)
    w0 = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    rows = [agent.train_epoch() for _ in range(2)]
    rn, v = agent.model.running_obs, agent.value_mean_std
    q.put((rank, w0, {k: x.detach().cpu().numpy() for k, x in agent.model.state_dict().items()}, rn.mean.cpu().numpy(), int(rn.n),
           [float(v.running_mean), float(v.running_var), float(v.count)], rows[-1]["step_rewards"], agent.experience_buffer.tensor_dict["rewards"].cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()
    task.close()


def test_two_ranks_train_epoch_keeps_the_replicas_identical():
    """PPOAgent.train_epoch on two ranks (each its own envs, seed 7 + rank): advantage statistics, observation / value normalisers and
    gradients are exchanged at the update, so after two epochs both replicas hold the same weights and normalisers, bit for bit,
    although they rolled out different envs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_ppo, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = res
    assert not np.array_equal(a[7], b[7]), "the ranks roll out different envs"
    moved = 0.0
    for k in a[2]:
        assert np.array_equal(a[1][k], b[1][k]), "same initial weights: " + k
        assert np.array_equal(a[2][k], b[2][k]), "replicas diverged: " + k
        moved = max(moved, float(np.abs(a[2][k] - a[1][k]).max()))
    assert moved > 1e-3
    assert np.array_equal(a[3], b[3]) and a[4] == b[4] == 2 * 2 * 3 * 2 * 32 * 32  # epochs x mini-epochs x minibatches x ranks x envs x steps
    assert a[5] == b[5]
    assert np.isfinite(a[6]) and np.isfinite(b[6])


def test_bench_runs_its_collective_path_under_rccl():
    """bench.py with V2P_BENCH_FORCE_DIST=1: RCCL process group of world size 1 on this GPU, the barrier, the all-gather of the rank times
    and the print-last logic of the N > 1 path execute for real (the driver's 8-GPU run takes exactly this path)."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, V2P_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "32", "--warmup", "32", "--num-envs", "1024", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["backend"] == "nccl(rccl)" and line["config"]["world_size_seen"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 1e5 and len(line["config"]["per_rank_env_steps_per_s"]) == 1
