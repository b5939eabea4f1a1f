"""Env sharding and the PPO-update collectives (one process per GPU, `torch.distributed`; backend
"nccl" is RCCL on ROCm, "gloo" in the CPU tests).

The rollout itself has NO collective: envs are independent, each rank owns its own
envs and a replica of the motion tables.  The reference scales with Horovod and normalises advantages
rank-locally (embodied_pose/agents/im_agent.py:461-473); the north star adds a global exchange at the
PPO update, provided here in the two forms SURVEY 2a discusses:

  all_gather_advantages   all-gather of advantages[T, N_local] / returns (1 MiB per rank at 32 x 8192;
                          one direct all-gather over the 7 xGMI links, ~7 us of wire time)
  global_advantage_stats  all-reduce of (sum, sum of squares, count): 3 numbers, sufficient for the mean/std
                          that `_calc_advs` needs
"""
import torch
import torch.distributed as dist


def shard_envs(num_envs_total, rank, world_size):
    """Contiguous shard [lo, hi) of rank `rank` (rank r owns envs r*N/W .. (r+1)*N/W)."""
    base, rem = divmod(num_envs_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_advantages(adv, group=None):
    """[T, N_local, ...] on every rank -> [T, N_global, ...] (envs concatenated in rank order).  Ranks may own different
    numbers of envs (shard_envs hands out uneven shards when N % W != 0): the per-rank counts are exchanged first and the
    payload is padded to the largest shard (one collective of equal-sized buffers), then trimmed."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return adv
    world = dist.get_world_size(group)
    local = adv.transpose(0, 1).contiguous()  # env-major so that the gather concatenates envs
    counts = torch.zeros(world, dtype=torch.int64, device=adv.device)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=adv.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts = [int(c) for c in counts.tolist()]
    cap = max(counts)
    if local.shape[0] < cap:
        pad = torch.zeros((cap - local.shape[0],) + tuple(local.shape[1:]), dtype=adv.dtype, device=adv.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=adv.dtype, device=adv.device)
    dist.all_gather_into_tensor(out, local, group=group)
    if all(c == cap for c in counts):
        return out.transpose(0, 1).contiguous()
    parts = [out[r * cap:r * cap + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0).transpose(0, 1).contiguous()


def global_advantage_stats(adv, mask=None, group=None):
    """Global masked mean / std of the advantages from a 3-number all-reduce.  The std is the unbiased one
    (torch's default, which `valid_advs.std()` in im_agent.py:470 uses)."""
    a = adv.double()
    m = torch.ones_like(a) if mask is None else mask.double().expand_as(a)
    s = torch.stack([(a * m).sum(), (a * a * m).sum(), m.sum()])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    mean = s[0] / s[2]
    var = torch.clamp((s[1] - s[2] * mean * mean) / torch.clamp(s[2] - 1.0, min=1.0), min=0.0)
    return mean.to(adv.dtype), var.sqrt().to(adv.dtype), s[2]


def normalize_advantages(adv, mask=None, group=None, eps=1e-8):
    """`_calc_advs` (im_agent.py:461-473) with global statistics."""
    mean, std, _ = global_advantage_stats(adv, mask, group)
    return (adv - mean) / (std + eps)
