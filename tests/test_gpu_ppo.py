"""Device-resident PPO rollout (vid2player3d_amd/ppo.py): the experience buffer of one 32-step epoch against a numpy
replay of the same steps - task ops through the task oracle teacher-forced with the recorded states, the bookkeeping (GAE, returns,
alive mask, episode statistics) recomputed in numpy the way im_agent.play_steps does it with host-side indexing."""
import numpy as np
import pytest
import torch

from oracle import task_oracle as O
from tests.gpu_util import DEV, N, make_task

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from vid2player3d_amd import motion_tables, synth
    from vid2player3d_amd.model import load_baked_model
    from vid2player3d_amd.motion_lib import MotionLib

    bm = load_baked_model()
    tabs = motion_tables.build_tables(synth.make_clips(11, 8, 90, 160), bm.parents, bm.local_pos)
    return bm, tabs, MotionLib(tabs, DEV)


def test_experience_buffer_matches_numpy_replay(setup):
    from vid2player3d_amd.ppo import PPOAgent

    bm, tabs, lib = setup
    n, T = 96, 32
    task = make_task(n, lib)
    agent = PPOAgent(task, horizon_length=T, seed=3, sigma_init=-0.5)  # wide action noise: some humanoids terminate inside the epoch
    torch.manual_seed(5)  # task.reset() draws the RSI phases from torch's global generator
    batch = agent.play_steps()
    torch.cuda.synchronize()
    td = {k: N(v) for k, v in agent.experience_buffer.tensor_dict.items()}
    times = N(task._reset_ref_motion_times)
    # ---- env side: task oracle, teacher-forced with the states recorded in the buffer (obs = the packed state, humanoid_smpl_im.py:198)
    ref = O.TaskOracle(tabs, N(task._reset_ref_motion_ids), bm.kp.astype(np.float32), term_heights=N(task._termination_heights))
    ref.reset_all(times)
    assert np.abs(td["obses"][0] - ref.obs_buf).max() < 5e-6
    died = 0
    for k in range(T):
        assert np.abs(td["obses"][k] - ref.obs_buf).max() < 5e-6, k
        ref.pre_physics_step(td["actions"][k].copy())
        o = td["next_obses"][k]
        rb = np.concatenate([o[:, 0:72].reshape(n, 24, 3), o[:, 72:168].reshape(n, 24, 4), o[:, 306:378].reshape(n, 24, 3), o[:, 378:450].reshape(n, 24, 3)], axis=-1)
        ref.set_sim_state(o[:, 168:237], o[:, 237:306], rb)
        ref.post_physics_step()
        assert np.abs(td["next_obses"][k] - ref.obs_buf).max() < 5e-6
        assert np.abs(td["rewards"][k, :, 0] - ref.rew_buf).max() < 2e-4, k
        assert np.array_equal(td["dones"][k], ref.reset_buf.astype(np.float32)), k
        died = int(ref.reset_buf.sum())
    assert 3 <= died < n, died
    # ---- bookkeeping, the reference's way (im_agent.py:389-407)
    adv = O.discount_values(td["dones"], td["values"], td["rewards"], td["next_values"], 0.99, 0.95)
    assert np.abs(N(batch["returns"]).transpose(1, 0, 2) - (adv + td["values"])).max() < 1e-5
    assert np.array_equal(N(batch["alive"]).T, 1.0 - td["dones"])
    # next_values: critic of the next obs, zeroed where the env terminated (end_value_type 'next'); here only the masking is checked
    term_steps = (np.diff(np.concatenate([np.zeros((1, n)), td["dones"]], 0), axis=0) > 0)
    assert (td["next_values"][:, :, 0][term_steps & (td["dones"] == 1)] == 0).mean() > 0.5
    # episode statistics accumulated on the device == host-side indexing of the reference (.nonzero(), game_rewards / game_lengths)
    acc, sub = [x.cpu().numpy() for x in batch["stats"]]
    prev = np.concatenate([np.zeros((1, n)), td["dones"][:-1]], 0)
    cur_r = np.cumsum(td["rewards"][:, :, 0], axis=0)
    step_done = td["dones"] * (1 - prev)
    fin_n = step_done.sum() + (1 - td["dones"][-1]).sum()
    fin_r = (cur_r * step_done).sum() + (cur_r[-1] * (1 - td["dones"][-1])).sum()
    fin_l = ((np.arange(T)[:, None] + 1) * step_done).sum() + (T * (1 - td["dones"][-1])).sum()
    assert abs(acc[0] - fin_n) < 1e-6 and abs(acc[1] - fin_r) < 1e-2 and abs(acc[2] - fin_l) < 1e-6
    assert abs(acc[3] - (1 - prev).sum()) < 1e-6 and abs(acc[4] - (td["rewards"][:, :, 0] * (1 - prev)).sum()) < 1e-2
    task.close()


def test_train_epoch_runs_and_improves_nothing_silly(setup):
    """Two PPO epochs on a small batch: finite losses, parameters move, the meters are consistent."""
    from vid2player3d_amd.ppo import PPOAgent

    _, _, lib = setup
    task = make_task(128, lib)
    agent = PPOAgent(task, minibatch_envs=64, mini_epochs=2, seed=1)
    w0 = [p.detach().clone() for p in agent.actor.parameters()]
    for _ in range(2):
        r = agent.train_epoch()
        assert np.isfinite([r["a_loss"], r["c_loss"], r["kl"], r["step_rewards"]]).all()
        assert r["frames"] == 128 * 32 and r["fps_total"] <= r["fps_step"]
        assert 0.0 < r["alive_ratio"] <= 1.0
    assert any(not torch.equal(a, b) for a, b in zip(w0, agent.actor.parameters()))
    assert "fps step" in agent.format_epoch_line(r)
    task.close()


def test_train_loop_saves_and_restores_reference_style_checkpoints(setup, tmp_path):
    """PPOAgent.train (ImitatorAgent.train, im_agent.py:164-269): epochs with the reference's log line, `<name>_latest.pth` in the reference's
    checkpoint layout; a second agent restored from it rolls out the same epoch."""
    from tests.test_ppo_reference import AMASS_IM_PARAMS
    from vid2player3d_amd.ppo import PPOAgent

    _, _, lib = setup
    task = make_task(128, lib)
    lines = []
    agent = PPOAgent.from_config(task, AMASS_IM_PARAMS, units=(64, 32), minibatch_envs=64, mini_epochs=2)
    agent.save_freq = 1
    r = agent.train(max_epochs=2, log=lines.append, network_path=str(tmp_path))
    assert agent.epoch_num == 2 and len(lines) == 2 and "fps step" in lines[0] and np.isfinite(r["step_rewards"])
    for name in ("Humanoid_latest.pth", "Humanoid_epoch00001.pth", "Humanoid_epoch00002.pth"):
        assert (tmp_path / name).exists(), name
    other = PPOAgent.from_config(task, AMASS_IM_PARAMS, units=(64, 32), minibatch_envs=64, mini_epochs=2)
    other.restore(str(tmp_path / "Humanoid_latest.pth"))
    assert other.epoch_num == 2 and other.frame == agent.frame
    for (k, x), (_, y) in zip(agent.model.state_dict().items(), other.model.state_dict().items()):
        assert torch.equal(x, y), k
    task.close()
