"""The PPO loop on the MI355X against the vectors recorded from the reference's own `ImitatorAgent` methods (oracle/gen_golden_ppo.py):
`play_steps` with the recorded rollout replayed as the env (network forward in eval mode through the HIP feature kernel, the policy-head
kernel - residual action, sample, neglogp -, the value normaliser, next-value masking, GAE kernel, alive mask, episode statistics), then
`prepare_dataset` + `calc_gradients` on the GPU with the HIP features.  And the two restructurings of the rollout that must not change a
bit: the critic evaluated once per step, and the envs split into two groups on two streams."""
import numpy as np
import pytest
import torch

from tests.gpu_util import DEV, N, make_task
from tests.test_ppo_reference import G, PAD, T, TR, check_update_against_golden, make_agent

pytestmark = pytest.mark.gpu


class TraceTask:
    """the recorded reference rollout (tests/golden/env_trace.npz, epoch 0) as a VecTask: whatever the actions, step k shows what the
    reference's task showed after its step k"""

    def __init__(self):
        t = lambda x, dt=torch.float32: torch.as_tensor(np.asarray(x)).to(device=DEV, dtype=dt).contiguous()  # noqa: E731
        self.device, self.num_envs, self.num_obs, self.num_actions, self.context_padding = DEV, 6, 461, 75, PAD
        self.context_feat = t(TR["e0_context_feat"])
        self.context_mask = t(TR["e0_context_mask"], torch.bool)
        self._obs = t(G["env/obs"])
        self._rew, self._done, self._term, self._sub = t(G["env/rewards"]), t(G["env/dones"], torch.long), t(G["env/terminate"], torch.long), t(G["env/sub_rewards"])
        self.obs_buf, self.rew_buf = torch.zeros((6, 461), device=DEV), torch.zeros(6, device=DEV)
        self.reset_buf = torch.zeros(6, dtype=torch.long, device=DEV)
        self.extras = {}
        self.k = 0

    def reset(self):
        self.k = 0
        self.obs_buf.copy_(self._obs[0])
        self.reset_buf.zero_()

    def step(self, actions):
        k = self.k
        self.obs_buf.copy_(self._obs[k + 1])
        self.rew_buf.copy_(self._rew[k])
        self.reset_buf.copy_(self._done[k])
        self.extras = {"terminate": self._term[k], "sub_rewards": self._sub[k]}
        self.k = k + 1


def test_play_steps_matches_the_references_method():
    agent = make_agent(TraceTask())
    mus, sig, act = (torch.as_tensor(G["play/" + k]).to(DEV) for k in ("mus", "sigmas", "actions"))
    agent.noise_fn = lambda n: ((act[:, n] - mus[:, n]) / sig[:, n]).contiguous()  # the draws the reference's Normal.sample() made
    batch = agent.play_steps()
    torch.cuda.synchronize()
    for k, tol in (("values", 2e-5), ("mus", 2e-5), ("sigmas", 1e-6), ("actions", 2e-5), ("neglogpacs", 2e-4), ("returns", 5e-5)):
        np.testing.assert_allclose(N(batch[k]), G["play/" + k], rtol=tol, atol=tol, err_msg=k)
    np.testing.assert_allclose(N(batch["next_values"]), G["play/next_values"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(N(batch["rewards"]), G["play/rewards"], rtol=0, atol=0)
    assert np.array_equal(N(batch["dones"]), G["play/dones"].astype(np.float32)) and np.array_equal(N(batch["alive"]), G["play/alive"])
    assert np.array_equal(N(batch["obses"]), G["play/obses"]) and np.array_equal(N(batch["next_obses"]), G["play/next_obses"])
    assert batch["played_frames"] == int(G["play/played_frames"])
    # the residual action is in the means: without it they would be off by the context's target DOF positions (O(1) rad)
    tgt = TR["e0_context_feat"][:, PAD:PAD + T, 168:237]
    assert np.abs(tgt).max() > 0.3
    # episode statistics: device accumulators == the reference's host-side meters
    acc, sub = (x.cpu().numpy() for x in batch["stats"])
    assert acc[0] == len(G["play/game_rewards"])
    np.testing.assert_allclose(acc[1], G["play/game_rewards"].sum(), rtol=1e-5)
    np.testing.assert_allclose(acc[2], G["play/game_lengths"].sum(), rtol=0)
    assert acc[3] == float(G["play/step_count"])
    np.testing.assert_allclose(acc[4] / acc[3], float(G["play/step_rewards_avg"][0]), rtol=1e-5)
    np.testing.assert_allclose(sub / acc[3], G["play/step_sub_rewards_avg"], rtol=1e-5)
    np.testing.assert_allclose(float(batch["alive"].mean()), float(G["play/alive_ratio"]), rtol=1e-6)


def test_update_on_the_gpu_matches_the_references_methods():
    """prepare_dataset (HIP feature kernel, training flavour) + calc_gradients on cuda tensors"""
    agent = make_agent(TraceTask())
    check_update_against_golden(agent, None, device=DEV, tol=2.0)


@pytest.fixture(scope="module")
def lib():
    from vid2player3d_amd import motion_tables, synth
    from vid2player3d_amd.model import load_baked_model
    from vid2player3d_amd.motion_lib import MotionLib

    bm = load_baked_model()
    tabs = motion_tables.build_tables(synth.make_clips(11, 8, 90, 160), bm.parents, bm.local_pos)
    return MotionLib(tabs, DEV)


def _rollout(lib, n, groups, reuse, seed=9, flat_heads=False, overlap=True):
    from vid2player3d_amd.ppo import PPOAgent

    rng = np.random.default_rng(seed)
    ids = rng.integers(0, 8, size=n)
    times = torch.as_tensor(rng.uniform(0.05, 1.2, size=n).astype(np.float32)).to(DEV)
    bounds = np.linspace(0, n, groups + 1).astype(int)
    tasks = [make_task(int(bounds[g + 1] - bounds[g]), lib, motion_ids=ids[bounds[g]:bounds[g + 1]]) for g in range(groups)]
    agent = PPOAgent(tasks if groups > 1 else tasks[0], units=(64, 32), seed=3, sigma_init=-0.7, reuse_next_values=reuse, overlap_critic=overlap)
    agent.model.running_obs.update(torch.randn(512, 734, device=DEV) * 2.0 + 0.3)  # (a normaliser that does something)
    if flat_heads:
        # the output layers answer with their bias whatever the input: a GEMM library may sum a row in a different order when the batch
        # has another number of rows (96 vs 192), and one ulp in an action is a different trajectory 32 contact-rich steps later
        with torch.no_grad():
            agent.model.mu.weight.zero_()
            agent.model.value.weight.zero_()
            agent.model.mu.bias.copy_(torch.linspace(-0.1, 0.1, 75))
    batch = agent.play_steps(reset_fn=lambda task, g: task.reset_with_times(None, times[bounds[g]:bounds[g + 1]].contiguous()))
    torch.cuda.synchronize()
    out = {k: N(v).copy() for k, v in agent.experience_buffer.tensor_dict.items()}
    out["returns"] = N(batch["returns"]).copy()
    out["stats"] = np.concatenate([N(x) for x in batch["stats"]])
    out["context"] = N(batch["context_feat"]).copy()
    for t in tasks:
        t.close()
    return out


def test_one_critic_pass_per_step_beside_the_physics_is_bit_identical(lib):
    """the reference's two critic passes per step == one pass whose result serves both == that pass on a side stream next to the physics"""
    a, b, c = _rollout(lib, 192, 1, False), _rollout(lib, 192, 1, True, overlap=False), _rollout(lib, 192, 1, True, overlap=True)
    assert a["dones"].sum() > 0 and np.abs(a["values"]).max() > 0 and np.abs(a["next_values"]).max() > 0
    for k in a:
        if k == "stats":  # (a: the torch bookkeeping, float32 sums; b, c: v2p_rollout_record, float64 block sums)
            np.testing.assert_allclose(a[k], b[k], rtol=1e-6)
            assert np.array_equal(b[k], c[k]) or np.allclose(b[k], c[k], rtol=1e-12)
            continue
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], c[k]), k


def test_fused_step_record_equals_the_torch_bookkeeping(lib):
    """v2p_rollout_record (one launch per step: buffer rows, dones / terminate as floats, episode returns / lengths, statistics) against
    the ~25 torch ops it replaces, on the same rollout: buffer bit for bit, statistics to float32 summation order; and on random inputs
    with an odd env count (scalar copy path, partial block)."""
    from vid2player3d_amd import _lib as L
    from vid2player3d_amd.ppo import PPOAgent

    outs = []
    for fused in (True, False):
        orig = PPOAgent.__init__

        def patched(self, *a, **kw):
            orig(self, *a, **kw)
            self.fused_record = fused
        PPOAgent.__init__ = patched
        try:
            outs.append(_rollout(lib, 192, 1, True))
        finally:
            PPOAgent.__init__ = orig
    a, b = outs
    assert a["dones"].sum() > 0
    for k in a:
        if k == "stats":
            np.testing.assert_allclose(a[k], b[k], rtol=1e-6)
        else:
            assert np.array_equal(a[k], b[k]), k
    n, D = 1001, 461
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    R = lambda *s: torch.rand(s, device=DEV, generator=g)  # noqa: E731
    obs, rew, sub = R(n, D), R(n), R(n, 4)
    reset, term = (R(n) < 0.3).long(), (R(n) < 0.1).long()
    prev, cr, cl = (R(n) < 0.2).float(), R(n) * 5, torch.floor(R(n) * 20)
    acc, sacc = torch.zeros(8, dtype=torch.float64, device=DEV), torch.zeros(4, dtype=torch.float64, device=DEV)
    big = torch.zeros((3, n + 7, D), device=DEV)  # an odd row offset: the destination is not 16-byte aligned
    row = big[1, 3:3 + n]
    rr, dr, dn, tm = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros((n, 1), device=DEV)
    want_cr, want_cl = cr + rew, cl + 1
    d = reset.float()
    sd, alive = d * (1 - prev), 1 - prev
    want_acc = [sd.double().sum(), (want_cr * sd).double().sum(), (want_cl * sd).double().sum(), alive.double().sum(), (rew * alive).double().sum()]
    want_sub = (sub * alive[:, None]).double().sum(0)
    L.check(L.load().v2p_rollout_record(n, L.ptr(obs), D, L.ptr(rew), L.ptr(reset), L.ptr(term), L.ptr(sub), L.ptr(row), L.ptr(rr), L.ptr(dr), L.ptr(dn), L.ptr(tm),
                                        L.ptr(prev), L.ptr(cr), L.ptr(cl), L.ptr(acc), L.ptr(sacc), L.current_stream(torch.device(DEV))), "v2p_rollout_record")
    torch.cuda.synchronize()
    assert torch.equal(row, obs) and torch.equal(rr, rew) and torch.equal(dr, d) and torch.equal(dn, d) and torch.equal(tm[:, 0], term.float())
    assert torch.equal(prev, d) and torch.equal(cr, want_cr) and torch.equal(cl, want_cl)
    assert float(big[1, :3].abs().sum()) == 0 and float(big[1, 3 + n:].abs().sum()) == 0 and float(big[0].abs().sum()) == 0  # nothing beside the row
    np.testing.assert_allclose(N(acc[:5]), [float(x) for x in want_acc], rtol=1e-12)
    np.testing.assert_allclose(N(sacc), N(want_sub), rtol=1e-12)
    assert float(acc[5:].abs().sum()) == 0


def test_fused_record_is_taken_by_the_engines_task_and_refused_for_other_layouts(lib):
    """advisor r5: the fused bookkeeping reads raw pointers with HumanoidSMPLIM's layouts; a fresh engine task qualifies (before its first
    step too: `extras` is still empty then), a CUDA task with bool flags, int32 flags or five sub-rewards does not (torch path instead)."""
    import types

    from tests.gpu_util import make_task
    from vid2player3d_amd.ppo import PPOAgent

    task = make_task(8, lib)
    assert task.extras == {} and PPOAgent._fused_record_ok(task)
    task.reset_with_times(None, torch.full((8,), 0.3, device=DEV))
    task.step(torch.zeros((8, 75), device=DEV))
    assert PPOAgent._fused_record_ok(task)

    def stub(**kw):
        d = dict(num_envs=8, num_obs=461, obs_buf=torch.zeros((8, 461), device=DEV), rew_buf=torch.zeros(8, device=DEV),
                 reset_buf=torch.zeros(8, dtype=torch.int64, device=DEV),
                 extras={"terminate": torch.zeros(8, dtype=torch.int64, device=DEV), "sub_rewards": torch.zeros((8, 4), device=DEV)})
        d.update(kw)
        return types.SimpleNamespace(**d)

    assert PPOAgent._fused_record_ok(stub())
    assert not PPOAgent._fused_record_ok(stub(reset_buf=torch.zeros(8, dtype=torch.bool, device=DEV)))
    assert not PPOAgent._fused_record_ok(stub(extras={"terminate": torch.zeros(8, dtype=torch.int32, device=DEV), "sub_rewards": torch.zeros((8, 4), device=DEV)}))
    assert not PPOAgent._fused_record_ok(stub(extras={"terminate": torch.zeros(8, dtype=torch.int64, device=DEV), "sub_rewards": torch.zeros((8, 5), device=DEV)}))
    assert not PPOAgent._fused_record_ok(stub(obs_buf=torch.zeros((8, 922), device=DEV)[:, ::2]))
    assert not PPOAgent._fused_record_ok(stub(rew_buf=torch.zeros(8, dtype=torch.float64, device=DEV)))
    task.close()


def test_two_rollout_groups_fill_the_same_buffer(lib):
    """192 envs as one batch on one stream == 96 + 96 on two streams (same clips, same start times, same noise per env)"""
    a, b = _rollout(lib, 192, 1, True, flat_heads=True), _rollout(lib, 192, 2, True, flat_heads=True)
    assert a["dones"].sum() > 0 and np.abs(a["actions"]).max() > 0.5
    for k in a:
        if k == "stats":
            np.testing.assert_allclose(a[k], b[k], rtol=1e-6)  # (float32 partial sums taken in two halves)
        else:
            assert np.array_equal(a[k], b[k]), k
