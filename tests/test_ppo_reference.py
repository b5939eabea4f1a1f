"""The PPO update of vid2player3d_amd/ppo.py against vectors recorded by running the reference's OWN methods (oracle/gen_golden_ppo.py ->
tests/golden/ppo_trace.npz): `_calc_advs`, `prepare_dataset` (value normaliser in training mode) and four `calc_gradients` calls (losses,
masked KL, clip fraction, per-minibatch RunningNorm update, residual action in the training-mode forward, gradient-norm clip, Adam) on a
real recorded rollout of 6 envs x 32 steps.  Pure torch on the CPU: the one HIP op of the update (the raw 734-d features) is supplied by
the numpy oracle here and compared with the kernel in tests/test_gpu_ppo_reference.py.  Plus: two data-parallel ranks (gloo) == one
process on the concatenated batch.

WHAT IS PINNED TO THE REFERENCE AND WHAT IS NOT.  The generator ran the reference's own methods, but rl_games (1.1.4, third-party, absent)
was stood in for by oracle/ref_shim/rl_games_restated.py, written from its published source by the author of the product code.  Golden
keys whose VALUE passes through one of those helpers are therefore compared with a restatement of the same provenance - SELF-REFERENTIAL
as far as that helper's conventions go (SELF_REFERENTIAL_KEYS below; the helpers' mathematics is pinned separately, against independent
closed forms, in tests/test_rl_games_helpers.py).  Everything else in the file - buffer contents, returns, advantages, the RunningNorm
of the observations, the reference's loss formulas, Adam - is the reference's own arithmetic."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import task_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ppo_trace.npz"))
TR = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_trace.npz"))
N_ENV, T, PAD = 6, 32, 8

# golden key (prefix) -> the restated rl_games helper its value went through when it was recorded
SELF_REFERENTIAL_KEYS = {
    "vms1": "RunningMeanStd (value normaliser after prepare_dataset: prior count 1, unbiased batch variance, population merge)",
    "data/old_values": "RunningMeanStd.forward (normalised with the statistics above, epsilon inside the square root, clamp +-5)",
    "data/returns": "RunningMeanStd.forward",
    "grad/*/kl": "policy_kl(reduce=False) with its 1e-5 guards, averaged by the masked mean (divisor = mask.numel(): the reference's own form, im_agent.py:573)",
    "grad/*/actor_loss": "apply_masks (divisor = number of elements of the mask)", "grad/*/critic_loss": "apply_masks", "grad/*/entropy": "apply_masks",
    "grad/*/w/*": "the weights after Adam depend on the three masked losses above",
    "play/neglogpacs": "ModelA2CContinuousLogStd.neglogp", "data/old_logp_actions": "ModelA2CContinuousLogStd.neglogp",
}


def test_self_referential_keys_exist_in_the_fixture():
    """(the table above names real keys: a renamed fixture entry must not silently drop out of it)"""
    import fnmatch

    for pat in SELF_REFERENTIAL_KEYS:
        assert any(fnmatch.fnmatch(k, pat) for k in G.files), pat


def stub_task(n=N_ENV, device="cpu"):
    return types.SimpleNamespace(device=device, num_envs=n, num_obs=461, num_actions=75, context_padding=PAD)


def reference_weights(prefix="w0/"):
    return {k[len(prefix):]: torch.as_tensor(G[k]) for k in G.files if k.startswith(prefix)}


def oracle_features(obses, context_feat):
    """[N,T,461] + [N,L,378] -> raw [N,T,734] (preprocess_input, flatten=True: row (e, k) pairs with context frame PAD + k)"""
    n, t = obses.shape[:2]
    o = obses.reshape(n * t, 461)
    c = context_feat[:, PAD:PAD + t].reshape(n * t, 378)
    f = O.obs_imitation_734(o[:, 0:72].reshape(-1, 24, 3), o[:, 72:168].reshape(-1, 24, 4), c[:, 0:72].reshape(-1, 24, 3), c[:, 72:168].reshape(-1, 24, 4),
                            o[:, 168:237], o[:, 237:306], c[:, 168:237], o[:, 306:378].reshape(-1, 24, 3), o[:, 378:450].reshape(-1, 24, 3), o[:, 450:461])
    return f.reshape(n, t, 734).astype(np.float32)


def make_agent(task=None, **kw):
    from vid2player3d_amd.ppo import PPOAgent

    agent = PPOAgent(task or stub_task(), horizon_length=T, units=tuple(int(u) for u in G["units"]), minibatch_envs=3, learning_rate=float(G["grad/lr"]), **kw)
    agent.model.load_reference_state_dict(reference_weights())
    agent.value_mean_std.running_mean.fill_(float(G["vms0"][0]))
    agent.value_mean_std.running_var.fill_(float(G["vms0"][1]))
    agent.value_mean_std.count.fill_(float(G["vms0"][2]))
    return agent


def golden_batch(device="cpu"):
    t = lambda k: torch.as_tensor(G[k]).to(device)  # noqa: E731
    obses = np.stack([G["env/obs"][k] for k in range(T)], axis=1)  # [N,T,461]: the observation BEFORE step k
    return {"obses": torch.as_tensor(obses).to(device), "values": t("play/values"), "returns": t("play/returns"), "alive": t("play/alive"),
            "neglogpacs": t("play/neglogpacs"), "actions": t("play/actions"), "mus": t("play/mus"), "sigmas": t("play/sigmas"),
            "dones": t("play/dones").float(), "context_feat": torch.as_tensor(TR["e0_context_feat"]).to(device)}


def test_calc_advs_matches_the_references_method():
    agent = make_agent()
    b = golden_batch()
    np.testing.assert_allclose(agent._calc_advs(b).numpy(), G["advs/normalized"], rtol=2e-5, atol=2e-6)
    agent.normalize_advantage = False
    np.testing.assert_allclose(agent._calc_advs(b).numpy(), G["advs/raw"], rtol=1e-6, atol=1e-7)


def check_update_against_golden(agent, feat_raw, device="cpu", tol=1.0):
    """prepare_dataset + the four recorded calc_gradients calls; tol scales the tolerances (GPU GEMMs sum in another order)"""
    b = golden_batch(device)
    agent.set_train()
    ds = agent.prepare_dataset(b, feat_raw=feat_raw)
    for k in ("old_values", "returns", "advantages", "old_logp_actions"):
        np.testing.assert_allclose(ds[k].cpu().numpy(), G["data/" + k], rtol=2e-5 * tol, atol=5e-6 * tol, err_msg=k)
    v = agent.value_mean_std
    np.testing.assert_allclose([float(v.running_mean), float(v.running_var), float(v.count)], G["vms1"], rtol=1e-6)
    perms = G["grad/perms"]
    call = 0
    for perm in perms:
        for i in range(2):
            idx = torch.as_tensor(perm[3 * i:3 * i + 3]).to(device)
            agent.grad_norm = float(G["grad/%d/grad_norm_clip" % call])
            r = agent.calc_gradients({k: x[idx] for k, x in ds.items()})
            for k in ("actor_loss", "critic_loss", "entropy", "kl"):
                np.testing.assert_allclose(float(r[k]), float(G["grad/%d/%s" % (call, k)]), rtol=2e-4 * tol, atol=2e-6 * tol, err_msg="call %d %s" % (call, k))
            assert abs(float(r["actor_clip_frac"]) - float(G["grad/%d/actor_clip_frac" % call])) <= 2.1 / 96, call  # (a count over 96 samples)
            rn = agent.model.running_obs
            assert int(rn.n) == int(G["grad/%d/rn_n" % call])
            np.testing.assert_allclose(rn.mean.cpu().numpy(), G["grad/%d/rn_mean" % call], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(rn.std.cpu().numpy(), G["grad/%d/rn_std" % call], rtol=1e-5, atol=1e-6)
            sd = agent.model.state_dict()
            for k, w in sd.items():
                ref = G["grad/%d/w/a2c_network.%s" % (call, k)]
                # Adam's first steps move every weight by ~lr whatever the gradient's size: compare in units of the step
                assert np.abs(w.cpu().numpy() - ref).max() <= 0.02 * float(G["grad/lr"]) * (call + 1) * tol + 1e-7, "call %d weight %s" % (call, k)
            call += 1
    assert call == int(G["grad/calls"])


def test_prepare_dataset_and_calc_gradients_match_the_references_methods():
    agent = make_agent()
    feat = torch.as_tensor(oracle_features(golden_batch()["obses"].numpy(), TR["e0_context_feat"]))
    check_update_against_golden(agent, feat)
    # the four updates moved the weights by far more than the comparison allows (the test has teeth)
    w_first = G["grad/0/w/a2c_network.mu.weight"]
    assert np.abs(agent.model.mu.weight.detach().numpy() - w_first).max() > 20 * 0.02 * float(G["grad/lr"]) * 4


def test_masked_mean_divides_by_the_number_of_elements():
    from vid2player3d_amd.ppo import masked_mean

    x, m = torch.tensor([[1.0], [3.0], [5.0], [7.0]]), torch.tensor([[1.0], [0.0], [1.0], [0.0]])
    assert float(masked_mean(x, m)) == 1.5  # (1 + 5) / 4, like (kl_dist * alive).sum() / alive.numel() at im_agent.py:573


def test_reference_checkpoint_names_load():
    agent = make_agent()
    sd = reference_weights()
    assert set(k for k in sd if "running_obs" not in k) == set("a2c_network." + k for k in agent.model.state_dict())
    assert int(agent.model.running_obs.n) == 1000


# ---------------------------------------------------------------- two data-parallel ranks == one process on the concatenated batch
def _rank_main(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    b = golden_batch()
    feat = torch.as_tensor(oracle_features(b["obses"].numpy(), TR["e0_context_feat"]))
    lo, hi = rank * 3, rank * 3 + 3
    agent = make_agent(stub_task(3))
    shard = {k: (v[lo:hi] if k != "context_feat" else v[lo:hi]) for k, v in b.items()}
    agent.set_train()
    ds = agent.prepare_dataset(shard, feat_raw=feat[lo:hi])
    adv = ds["advantages"].clone()
    for order in ([0, 1, 2], [2, 0, 1]):
        for i in range(3):  # minibatches of one env per rank
            j = order[i]
            agent.minibatch_envs = 1
            agent.calc_gradients({k: x[j:j + 1] for k, x in ds.items()})
    q.put((rank, adv.numpy(), {k: v.numpy() for k, v in agent.model.state_dict().items()}, agent.model.running_obs.mean.numpy(),
           [float(agent.value_mean_std.running_mean), float(agent.value_mean_std.running_var), float(agent.value_mean_std.count)]))
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_concatenated_batch():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, 6 envs, minibatch k = {env order[k] of rank 0, env order[k] of rank 1}
    b = golden_batch()
    feat = torch.as_tensor(oracle_features(b["obses"].numpy(), TR["e0_context_feat"]))
    agent = make_agent()
    agent.set_train()
    ds = agent.prepare_dataset(b, feat_raw=feat)
    for order in ([0, 1, 2], [2, 0, 1]):
        for i in range(3):
            idx = torch.tensor([order[i], 3 + order[i]])
            agent.calc_gradients({k: x[idx] for k, x in ds.items()})
    adv_two = np.concatenate([got[0][1], got[1][1]], axis=0)
    np.testing.assert_allclose(adv_two, ds["advantages"].numpy(), rtol=1e-5, atol=1e-6)  # global advantage statistics
    for k, w in agent.model.state_dict().items():
        assert np.array_equal(got[0][2][k], got[1][2][k]), "ranks hold different weights: " + k
        assert np.abs(got[0][2][k] - w.numpy()).max() <= 0.02 * float(G["grad/lr"]) * 6 + 1e-7, k
    assert np.array_equal(got[0][3], got[1][3])
    np.testing.assert_allclose(got[0][3], agent.model.running_obs.mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[0][4], got[1][4])
    v = agent.value_mean_std
    np.testing.assert_allclose(got[0][4], [float(v.running_mean), float(v.running_var), float(v.count)], rtol=1e-9)


# ---------------------------------------------------------------- world 8: even and uneven env shards
def tiled_batch(m):
    """the golden 6-env rollout tiled to m envs (env e = golden env e % 6, shifted a little per tile so that no two envs are equal)"""
    b = golden_batch()
    idx = torch.arange(m) % N_ENV
    tile = (torch.arange(m) // N_ENV).float()
    out = {}
    for k, v in b.items():
        x = v[idx].clone()
        if k in ("obses", "values", "returns", "actions", "mus", "context_feat"):
            x = x + 0.01 * tile.reshape((m,) + (1,) * (x.dim() - 1))
        out[k] = x
    return out


def _feat_of(b):
    return torch.as_tensor(oracle_features(b["obses"].numpy(), b["context_feat"].numpy()))


def _rank8_main(rank, world, port, q, bounds, m):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    b = tiled_batch(m)
    lo, hi = bounds[rank], bounds[rank + 1]
    shard = {k: v[lo:hi] for k, v in b.items()}
    agent = make_agent(stub_task(hi - lo))
    agent.set_train()
    ds = agent.prepare_dataset(shard, feat_raw=_feat_of(shard))
    adv = ds["advantages"].clone()
    per = max(bounds[r + 1] - bounds[r] for r in range(world))
    for k in range(per):  # minibatch k = env k of every rank's shard (ranks with fewer envs wrap around: every rank joins every all-reduce)
        j = k % (hi - lo)
        agent.minibatch_envs = 1
        agent.calc_gradients({kk: x[j:j + 1] for kk, x in ds.items()})
    # the north star's all-gather of the advantages: every rank ends with the whole job's, in rank order
    from vid2player3d_amd import dist as vdist
    gathered = vdist.all_gather_advantages(adv.transpose(0, 1).contiguous())
    q.put((rank, adv.numpy(), {k: v.numpy() for k, v in agent.model.state_dict().items()}, agent.model.running_obs.mean.numpy(),
           [float(agent.value_mean_std.running_mean), float(agent.value_mean_std.running_var), float(agent.value_mean_std.count)], gathered.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("m,even", [(16, True), (12, False)])
def test_eight_ranks_hold_identical_replicas_and_global_statistics(m, even):
    """World 8 over gloo (BASELINE config 5's launch shape; the only multi-rank evidence short of an 8-GPU node): 16 envs as 8 x 2, and 12
    envs as the uneven shards shard_envs hands out (2,2,2,2,1,1,1,1).  Every rank ends with bit-identical weights and normalisers; the
    advantages are normalised with the statistics of the WHOLE job; the all-gather returns the whole job's advantages in rank order.  With
    even shards the replicas also equal ONE process on the concatenated batch (averaging per-rank means = the mean over the union)."""
    import torch.multiprocessing as mp

    from vid2player3d_amd.dist import shard_envs

    world = 8
    bounds = [shard_envs(m, r, world)[0] for r in range(world)] + [m]
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    assert (len(set(sizes)) == 1) == even and min(sizes) >= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000 + (0 if even else 7)
    procs = [ctx.Process(target=_rank8_main, args=(r, world, port, q, bounds, m)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(1, world):
        for k in got[0][2]:
            assert np.array_equal(got[0][2][k], got[r][2][k]), "rank %d holds different weights: %s" % (r, k)
        assert np.array_equal(got[0][3], got[r][3]) and got[0][4] == got[r][4]
        assert np.array_equal(got[0][5], got[r][5])
    # one process on the concatenated batch
    b = tiled_batch(m)
    agent = make_agent(stub_task(m))
    agent.set_train()
    ds = agent.prepare_dataset(b, feat_raw=_feat_of(b))
    adv_all = np.concatenate([g[1] for g in got], axis=0)
    np.testing.assert_allclose(adv_all, ds["advantages"].numpy(), rtol=1e-5, atol=1e-6)          # global advantage statistics
    np.testing.assert_allclose(got[0][5].transpose(1, 0), adv_all, rtol=0, atol=0)               # the all-gather: rank order, uneven shards trimmed
    v = agent.value_mean_std
    np.testing.assert_allclose(got[0][4], [float(v.running_mean), float(v.running_var), float(v.count)], rtol=1e-9)  # merged moments are exact for any shard sizes
    if even:
        for k in range(sizes[0]):
            idx = torch.tensor([bounds[r] + k for r in range(world)])
            agent.calc_gradients({kk: x[idx] for kk, x in ds.items()})
        for k, w in agent.model.state_dict().items():
            assert np.abs(got[0][2][k] - w.numpy()).max() <= 0.02 * float(G["grad/lr"]) * sizes[0] + 1e-7, k
        np.testing.assert_allclose(got[0][3], agent.model.running_obs.mean.numpy(), rtol=1e-5, atol=1e-6)


def _bcast_main(rank, world, port, q, ckpt):
    import torch.distributed as dist
    from vid2player3d_amd.ppo import PPOAgent

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(100 + rank)
    # DIFFERENT seeds per rank: equal replicas must come from the broadcast, not from equal seeds
    agent = PPOAgent(stub_task(3), horizon_length=T, units=(32, 16), minibatch_envs=3, seed=5 + 11 * rank)
    w_init = {k: v.clone().numpy() for k, v in agent.model.state_dict().items()}
    # a checkpoint only rank 0 can read (with Adam state): restore() hands it to every rank
    agent.restore(ckpt if rank == 0 else ckpt + ".not-on-this-rank")
    st = agent.optimizer.state_dict()["state"]
    q.put((rank, w_init, {k: v.clone().numpy() for k, v in agent.model.state_dict().items()}, agent.model.running_obs.mean.numpy().copy(),
           [float(agent.value_mean_std.running_mean), float(agent.value_mean_std.count)], agent.epoch_num, agent.frame,
           {k: {n: np.asarray(t) for n, t in v.items()} for k, v in st.items()}))
    dist.destroy_process_group()


def test_ranks_start_from_rank_0s_state(tmp_path):
    """PPOAgent.__init__ and restore() broadcast rank 0's model / normalisers / optimizer state / counters (the reference:
    hvd.setup_algo, im_agent.py:174-175): two ranks built with different seeds hold rank 0's weights, and a checkpoint that only rank 0
    can read continues identically on both."""
    import torch.multiprocessing as mp
    from vid2player3d_amd.ppo import PPOAgent

    src = PPOAgent(stub_task(3), horizon_length=T, units=(32, 16), minibatch_envs=3, seed=99)
    for p_ in src.model.parameters():  # one Adam step so that the checkpoint carries exp_avg / exp_avg_sq / step
        p_.grad = torch.full_like(p_, 0.01)
    src.optimizer.step()
    src.model.running_obs.mean.fill_(0.25)
    src.value_mean_std.running_mean.fill_(1.5)
    src.value_mean_std.count.fill_(77.0)
    src.epoch_num, src.frame = 12, 3456
    ckpt = src.save(str(tmp_path / "ck"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bcast_main, args=(r, 2, port, q, ckpt)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = got
    own1 = PPOAgent(stub_task(3), horizon_length=T, units=(32, 16), minibatch_envs=3, seed=5 + 11).model.state_dict()
    for k in r0[1]:
        assert np.array_equal(r0[1][k], r1[1][k]), "after __init__: " + k               # rank 1 holds rank 0's initial weights ...
    assert any(not np.array_equal(r1[1][k], own1[k].numpy()) for k in r1[1])            # ... not the ones its own seed gives
    want = src.model.state_dict()
    for k in r0[2]:
        assert np.array_equal(r0[2][k], want[k].numpy()) and np.array_equal(r1[2][k], want[k].numpy()), "after restore: " + k
    assert np.all(r1[3] == 0.25) and r1[4] == [1.5, 77.0] and (r1[5], r1[6]) == (12, 3456)
    assert len(r1[7]) == len(r0[7]) > 0
    for k in r0[7]:
        for n in ("exp_avg", "exp_avg_sq", "step"):
            assert np.array_equal(r0[7][k][n], r1[7][k][n]), (k, n)
        assert float(r1[7][k]["step"]) == 1.0


def _bcast_mismatch_main(rank, world, port, q):
    import torch.distributed as dist
    from vid2player3d_amd.ppo import PPOAgent

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    agent = PPOAgent(stub_task(3), horizon_length=T, units=(32, 16), minibatch_envs=3, seed=5)
    if rank == 1:  # Adam state on one rank only, and nobody materialises it on the other: the layouts differ
        for p_ in agent.model.parameters():
            p_.grad = torch.full_like(p_, 0.01)
        agent.optimizer.step()
    try:
        agent.broadcast_state()
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.destroy_process_group()


def test_broadcast_state_refuses_mismatched_layouts_on_every_rank():
    """advisor r5: ranks whose state differs in layout (optimizer slots on one rank only; `step` counters on different devices used to split
    the dtype groups differently) must not enter mismatched collectives - the layout manifest is exchanged first and EVERY rank raises."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bcast_mismatch_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("different state layouts" in got[r] for r in (0, 1)), got


# ---------------------------------------------------------------- the reference agent's configuration and checkpoint surface
AMASS_IM_PARAMS = {  # the `params` block of embodied_pose/cfg/amass_im.yaml:55-143, as yaml.safe_load returns it
    "seed": 7, "algo": {"name": "pose_im_rnn"}, "model": {"name": "pose_im"},
    "network": {"name": "pose_im_rnn", "separate": True, "use_running_obs": True, "running_obs_type": "ours", "context_padding": 8,
                "space": {"continuous": {"mu_activation": "None", "sigma_activation": "None", "mu_init": {"name": "default"},
                                         "sigma_init": {"name": "const_initializer", "val": -1.756}, "fixed_sigma": True, "learn_sigma": False}},
                "mlp": {"units": [1024, 1024, 512], "activation": "relu", "d2rl": False}},
    "config": {"name": "Humanoid", "ppo": True, "mixed_precision": False, "normalize_input": False, "normalize_value": True, "normalize_advantage": True,
               "gamma": 0.99, "tau": 0.95, "learning_rate": "2e-5", "lr_schedule": "constant", "max_epochs": 10000, "save_frequency": 200,
               "entropy_coef": 0.0, "truncate_grads": True, "grad_norm": 50.0, "e_clip": 0.2, "horizon_length": 32, "minibatch_size": 512,
               "mini_epochs": 6, "critic_coef": 5, "clip_value": False},
}


def test_agent_from_the_references_yaml_block():
    from vid2player3d_amd.ppo import PPOAgent

    params = AMASS_IM_PARAMS
    ref_yaml = "/root/reference/embodied_pose/cfg/amass_im.yaml"
    if os.path.exists(ref_yaml):  # (build container only) the literal above IS what the reference's file says
        import yaml

        with open(ref_yaml) as f:
            real = yaml.safe_load(f)["params"]
        for blk in ("config", "network"):
            for k, v in AMASS_IM_PARAMS[blk].items():
                if k not in ("mlp", "space"):
                    assert real[blk][k] == v or str(real[blk][k]) == str(v), (blk, k, real[blk][k], v)
        assert real["network"]["mlp"]["units"] == [1024, 1024, 512] and real["network"]["space"]["continuous"]["sigma_init"]["val"] == -1.756
        params = real
    a = PPOAgent.from_config(stub_task(8192), params, units=(32, 16))  # (small MLPs: the test only reads the hyper-parameters)
    assert (a.horizon_length, a.gamma, a.tau, a.last_lr, a.e_clip, a.critic_coef, a.mini_epochs, a.minibatch_envs, a.grad_norm) == (32, 0.99, 0.95, 2e-5, 0.2, 5, 6, 512, 50.0)
    assert a.normalize_value and a.normalize_advantage and not a.mixed_precision and a.max_epochs == 10000 and a.save_freq == 200
    assert float(a.model.sigma[0]) == pytest.approx(-1.756) and a.model.residual_action
    bad = {**AMASS_IM_PARAMS, "config": {**AMASS_IM_PARAMS["config"], "clip_value": True}}
    with pytest.raises(NotImplementedError):
        PPOAgent.from_config(stub_task(), bad)


def test_checkpoint_round_trip_with_the_references_names(tmp_path):
    agent = make_agent()
    feat = torch.as_tensor(oracle_features(golden_batch()["obses"].numpy(), TR["e0_context_feat"]))
    agent.set_train()
    ds = agent.prepare_dataset(golden_batch(), feat_raw=feat)
    agent.calc_gradients({k: v[:3] for k, v in ds.items()})  # (an optimizer with state, normalisers that have seen data)
    agent.epoch_num, agent.frame = 17, 17 * 192
    path = agent.save(str(tmp_path / "Humanoid_latest"))
    ck = torch.load(path, weights_only=False)
    # the key names of a reference checkpoint's `model` entry (tests/golden/ppo_trace.npz holds the reference module's own state_dict)
    assert set(ck["model"]) == set(reference_weights()) and set(ck["reward_mean_std"]) == {"running_mean", "running_var", "count"}
    other = make_agent()
    other.restore(path)
    for (k, x), (_, y) in zip(agent.model.state_dict().items(), other.model.state_dict().items()):
        assert torch.equal(x, y), k
    assert torch.equal(agent.model.running_obs.mean, other.model.running_obs.mean) and int(other.model.running_obs.n) == int(agent.model.running_obs.n)
    assert float(other.value_mean_std.count) == float(agent.value_mean_std.count) and (other.epoch_num, other.frame) == (17, 17 * 192)
    # both continue identically (the optimizer state came along)
    ra = agent.calc_gradients({k: v[3:] for k, v in ds.items()})
    other.set_train()
    other.dataset = ds
    rb = other.calc_gradients({k: v[3:] for k, v in ds.items()})
    assert float(ra["actor_loss"]) == float(rb["actor_loss"])
    for (k, x), (_, y) in zip(agent.model.state_dict().items(), other.model.state_dict().items()):
        assert torch.equal(x, y), k


def test_player_from_the_references_yaml_block():
    """players/im_player.py + rl_games' BasePlayer defaults [1.1.4, from memory]: 2000 games, deterministic, max_steps 27000"""
    from vid2player3d_amd.player import ImitatorPlayer

    task = stub_task(16)
    p = ImitatorPlayer.from_config(task, AMASS_IM_PARAMS, units=(32, 16))
    assert (p.games_num, p.is_determenistic, p.n_game_life, p.print_stats, p.max_steps, p.config_name) == (2000, True, 1, True, 27000, "Humanoid")
    assert float(p.model.sigma[0]) == pytest.approx(-1.756) and p.model.residual_action and not p.model.training
    with_player = {**AMASS_IM_PARAMS, "config": {**AMASS_IM_PARAMS["config"], "player": {"games_num": 5, "determenistic": False, "print_stats": False}}}
    p = ImitatorPlayer.from_config(task, with_player, units=(32, 16))
    assert (p.games_num, p.is_determenistic, p.print_stats) == (5, False, False)
    with pytest.raises(NotImplementedError):
        ImitatorPlayer.from_config(task, {**AMASS_IM_PARAMS, "config": {**AMASS_IM_PARAMS["config"], "normalize_input": True}})
    # a checkpoint of the training agent carries what the player loads (im_player.py:43-51: `model`, normalisers inside it)
    agent = make_agent()
    q = ImitatorPlayer(stub_task(), units=tuple(int(u) for u in G["units"]))
    q.set_weights(agent.get_full_state_weights())
    for (k, x), (_, y) in zip(agent.model.state_dict().items(), q.model.state_dict().items()):
        assert torch.equal(x, y), k
    assert float(q.value_mean_std.running_mean) == float(agent.value_mean_std.running_mean)
