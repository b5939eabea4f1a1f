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
