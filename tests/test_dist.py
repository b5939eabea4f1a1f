"""N>1 path on CPU: env sharding + the PPO-update collectives with gloo, world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vid2player3d_amd.dist import all_gather_advantages, global_advantage_stats, normalize_advantages, shard_envs


def test_shard_envs_partitions():
    for total, world in ((65536, 8), (10, 3), (7, 8)):
        spans = [shard_envs(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


TOTAL_ENVS = 13  # odd on purpose: shard_envs gives the two ranks 7 and 6 envs, the gather must cope with uneven shards


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(32 * TOTAL_ENVS, dtype=torch.float32).reshape(32, TOTAL_ENVS) * 0.37 - 3.0
    mask_full = (torch.arange(32 * TOTAL_ENVS).reshape(32, TOTAL_ENVS) % 5 != 0).float()
    lo, hi = shard_envs(TOTAL_ENVS, rank, world)
    adv, mask = full[:, lo:hi].clone(), mask_full[:, lo:hi].clone()
    gathered = all_gather_advantages(adv)
    mean, std, cnt = global_advantage_stats(adv, mask)
    norm = normalize_advantages(adv, mask)
    q.put((rank, gathered.numpy(), float(mean), float(std), float(cnt), norm.numpy(), lo, hi))
    dist.destroy_process_group()


def test_collectives_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = (np.arange(32 * TOTAL_ENVS, dtype=np.float32).reshape(32, TOTAL_ENVS) * 0.37 - 3.0)
    mask = (np.arange(32 * TOTAL_ENVS).reshape(32, TOTAL_ENVS) % 5 != 0).astype(np.float64)
    mean = (full * mask).sum() / mask.sum()
    std = np.sqrt(((full.astype(np.float64) - mean) ** 2 * mask).sum() / (mask.sum() - 1))  # unbiased, like torch.std
    assert sorted(hi - lo for _, _, _, _, _, _, lo, hi in res) == [6, 7]
    for rank, gathered, m, s, c, norm, lo, hi in res:
        assert np.array_equal(gathered, full)
        assert abs(m - mean) < 1e-5 and abs(s - std) < 1e-5 and c == mask.sum()
        assert np.allclose(norm, (full[:, lo:hi] - mean) / (std + 1e-8), atol=1e-5)
