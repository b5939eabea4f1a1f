"""The evaluation rollout (vid2player3d_amd/player.py = players/im_player.py of the reference, `run.py --test`) on the MI355X:

  * against vectors recorded from the reference's own `ImitatorPlayer.get_action / env_step / run` (oracle/gen_golden_player.py): the
    recorded trace replayed as the env for 40 steps - past the 32-step context window, so the window is rebuilt once -, deterministic
    actions (residual action included) step by step, the finished-episode bookkeeping and both exits of the loop;
  * `task._init_context(ids, times)` alone (`v2p_env_context`): the window rebuilt around the CURRENT clip times of a running rollout,
    against the task oracle, nothing else touched;
  * end to end on the engine: a rollout of 70 steps without a reset, from a checkpoint the training agent wrote."""
import os

import numpy as np
import pytest
import torch

from oracle import task_oracle as O
from tests.gpu_util import DEV, N, make_task
from tests.test_ppo_reference import G, PAD, reference_weights

pytestmark = pytest.mark.gpu
P = np.load(os.path.join(os.path.dirname(__file__), "golden", "player_trace.npz"))


class TraceTask:
    """the recorded rollout of the generator as a task: step k shows what the reference's task showed after its step k; `_init_context`
    switches to the window the reference's task built for its second epoch"""

    context_length = 32

    def __init__(self, perm):
        t = lambda x, dt=torch.float32: torch.as_tensor(np.asarray(x)).to(device=DEV, dtype=dt).contiguous()  # noqa: E731
        self.device, self.num_envs, self.num_obs, self.num_actions, self.context_padding = DEV, 6, 461, 75, PAD
        perm = np.asarray(perm)
        self._obs, self._rew, self._done = t(P["env/obs"][:, perm]), t(P["env/rewards"][:, perm]), t(P["env/dones"][:, perm], torch.long)
        self._windows = t(P["env/context_feat"][:, perm])
        self.context_feat = self._windows[0].clone()
        self._reset_ref_motion_ids = torch.arange(6, device=DEV)
        self._cur_ref_motion_times = torch.zeros(6, device=DEV)
        self.obs_buf, self.rew_buf = torch.zeros((6, 461), device=DEV), torch.zeros(6, device=DEV)
        self.reset_buf = torch.zeros(6, dtype=torch.long, device=DEV)
        self.extras = {}
        self.k, self.resets, self.context_calls, self.actions = 0, 0, [], []

    def render_vis(self, init=False):
        pass

    def reset(self, env_ids=None):
        assert env_ids is None
        self.k = 0
        self.resets += 1
        self.context_feat.copy_(self._windows[0])
        self.obs_buf.copy_(self._obs[0])
        self.reset_buf.zero_()

    def _init_context(self, motion_ids, motion_times):
        assert motion_ids is self._reset_ref_motion_ids and motion_times is self._cur_ref_motion_times
        self.context_calls.append(self.k)
        self.context_feat.copy_(self._windows[1])

    def step(self, actions):
        k = self.k
        self.actions.append(actions.clone())
        self.obs_buf.copy_(self._obs[k + 1])
        self.rew_buf.copy_(self._rew[k])
        self.reset_buf.copy_(self._done[k])
        self.k = k + 1


@pytest.mark.parametrize("tag", ["a", "b"])
def test_run_matches_the_references_player(tag):
    from vid2player3d_amd.player import ImitatorPlayer

    task = TraceTask(P[tag + "/perm"])
    lines = []
    player = ImitatorPlayer(task, units=tuple(int(u) for u in G["units"]), games_num=int(P[tag + "/games_num"]), deterministic=True,
                            max_steps=int(P["num_steps"]), log=lambda *a: lines.append(" ".join(str(x) for x in a)))
    player.model.load_reference_state_dict(reference_weights())
    res = player.run()
    want = P[tag + "/actions"]
    assert len(task.actions) == want.shape[0] and task.resets == int(P[tag + "/resets"]) and task.context_calls == P[tag + "/context_calls"].tolist()
    got = np.stack([N(a) for a in task.actions])
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    assert np.abs(want[:, :, :69]).max() > 0.3  # (the residual action is in them)
    if tag == "a":  # steps 32.. are computed against the rebuilt window: with the old one they would be elsewhere
        assert want.shape[0] > 33 and np.abs(P["env/context_feat"][1][:, PAD:PAD + 3, 168:237] - P["env/context_feat"][0][:, PAD:PAD + 3, 168:237]).max() > 0.05
    np.testing.assert_allclose(res["sum_rewards"], float(P[tag + "/sum_rewards"]), rtol=1e-5)
    np.testing.assert_allclose(res["av_reward"], float(P[tag + "/av_reward"]), rtol=1e-5)
    np.testing.assert_allclose(res["av_steps"], float(P[tag + "/av_steps"]), rtol=0)
    per = np.asarray([[float(ln.split()[1]), float(ln.split()[3])] for ln in lines if ln.startswith("reward:")])
    np.testing.assert_allclose(per, P[tag + "/per_step_stats"], rtol=1e-5)
    assert res["rounds"] == task.resets


@pytest.fixture(scope="module")
def setup():
    from vid2player3d_amd import motion_tables, synth
    from vid2player3d_amd.model import load_baked_model
    from vid2player3d_amd.motion_lib import MotionLib

    bm = load_baked_model()
    tabs = motion_tables.build_tables(synth.make_clips(11, 8, 90, 160), bm.parents, bm.local_pos)
    return bm, tabs, MotionLib(tabs, DEV)


def test_init_context_alone_rebuilds_the_window_at_the_current_times(setup):
    """humanoid_smpl_im.py:530-563 called from outside a reset (players/im_player.py:238-240)"""
    bm, tabs, lib = setup
    n = 64
    task = make_task(n, lib)
    torch.manual_seed(2)
    task.reset()
    first = N(task.context_feat).copy()
    for _ in range(5):
        task.step((0.05 * torch.randn(n, 75, device=DEV)).contiguous())
    before = {k: N(getattr(task, k)).copy() for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "_cur_ref_motion_times", "_dof_pos", "_rigid_body_pos")}
    times = N(task._cur_ref_motion_times)
    assert np.abs(times - N(task._reset_ref_motion_times) - 5 * task.dt).max() < 1e-5
    task._init_context(task._reset_ref_motion_ids, task._cur_ref_motion_times)
    torch.cuda.synchronize()
    feat, mask = O.init_context(tabs, N(task._reset_ref_motion_ids), times, task.dt, task.context_length, task.context_padding)
    got = N(task.context_feat)
    assert np.abs(got - feat.reshape(got.shape)).max() < 5e-6
    assert np.array_equal(N(task.context_mask), mask.reshape(n, -1))
    assert np.abs(got - first).max() > 1e-3  # (five steps later: another window; frame w of the new one = frame w + 5 of the old one)
    assert np.abs(got[:, :-5] - first[:, 5:]).max() < 2e-5
    for k, v in before.items():
        assert np.array_equal(N(getattr(task, k)), v), k
    # a copy of the ids is accepted, other ids and a partial list are not
    task._init_context(task._reset_ref_motion_ids.clone(), task._cur_ref_motion_times)
    with pytest.raises(RuntimeError):
        task._init_context((task._reset_ref_motion_ids + 1) % 8, task._cur_ref_motion_times)
    with pytest.raises(RuntimeError):
        task._init_context(task._reset_ref_motion_ids, task._cur_ref_motion_times[:10])
    task.close()


def test_player_runs_past_the_context_window_from_a_training_checkpoint(setup, tmp_path):
    from tests.test_ppo_reference import AMASS_IM_PARAMS
    from vid2player3d_amd.player import ImitatorPlayer
    from vid2player3d_amd.ppo import PPOAgent

    _, _, lib = setup
    n = 96
    task = make_task(n, lib, stateInit="Start", episodeLength=80)
    agent = PPOAgent.from_config(task, AMASS_IM_PARAMS, units=(64, 32), minibatch_envs=48, mini_epochs=1)
    agent.train_epoch()
    path = agent.save(str(tmp_path / "Humanoid_latest"))
    calls = []
    inner = task._init_context
    task._init_context = lambda ids, times: (calls.append(float(times[0])), inner(ids, times))[1]
    player = ImitatorPlayer.from_config(task, AMASS_IM_PARAMS, units=(64, 32), games_num=n, max_steps=70, network_path=str(tmp_path), log=None)
    player.restore("latest")
    for (k, x), (_, y) in zip(agent.model.state_dict().items(), player.model.state_dict().items()):
        assert torch.equal(x, y), k
    assert torch.equal(agent.model.running_obs.mean, player.model.running_obs.mean) and int(player.model.running_obs.n) > 0
    res = player.run()
    assert os.path.basename(path) == "Humanoid_latest.pth"
    # stateInit Start: every clip from t = 0; env 0 decides when a round ends (done[0]); the window is rebuilt every 32 steps of a round
    assert res["rounds"] >= 1 and res["games_played"] >= 1 and np.isfinite([res["av_reward"], res["av_steps"]]).all()
    assert 0 < res["av_steps"] <= 80 and res["av_reward"] > 0
    steps_per_round = res["env_steps"] // n / res["rounds"]
    if steps_per_round > 32:
        assert len(calls) >= 1 and abs(calls[0] - 32 * task.dt) < 1e-4
    # the player over the agent's own network: same actions as the restored one
    shared = ImitatorPlayer.from_agent(agent, games_num=1, max_steps=3, log=None)
    task.reset()
    obs = {"obs": task.obs_buf, "t": 0}
    assert torch.equal(shared.get_action(obs, True), player.get_action(obs, True))
    sampled = shared.get_action(obs, False)
    assert not torch.equal(sampled, shared.get_action(obs, True)) and torch.isfinite(sampled).all()
    vals = shared.get_action_values(obs, 0)
    assert vals["values"].shape == (n, 1) and vals["neglogpacs"].shape == (n,)
    task.close()
