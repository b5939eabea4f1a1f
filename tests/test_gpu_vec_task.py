"""The surface rl_games sees (vec_task.py:16-63, 120-138; vec_task_wrappers.py:22-28; run.py:93-137) over the HIP engine."""
import numpy as np
import pytest
import torch

from tests.gpu_util import DEV, N, make_task, synth_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mlib():
    from vid2player3d_amd.motion_lib import MotionLib

    return MotionLib(synth_tables(seed=9, num_clips=4, min_frames=60, max_frames=90), DEV)


def test_rlgpu_env_stack(mlib):
    from vid2player3d_amd.vec_task import RLGPUEnv, VecTaskPythonWrapper

    n = 40
    task = make_task(n, mlib)
    env = RLGPUEnv(VecTaskPythonWrapper(task, DEV, clip_observations=5.0, clip_actions=float("inf")))
    info = env.get_env_info()
    assert info["observation_space"].shape == (461,) and info["action_space"].shape == (75,) and env.get_number_of_agents() == 1
    obs0 = env.reset()
    assert obs0.shape == (n, 461) and float(obs0.abs().max()) <= 5.0
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    task.reset_buf[3] = 1  # a dead env: the TASK's copy of the actions is masked, the policy's tensor is not
    pol = torch.cat([task._target_dof_pos + 0.1 * torch.randn((n, 69), device=DEV, generator=g), 0.1 * torch.randn((n, 6), device=DEV, generator=g)], dim=1)
    keep = pol.clone()
    obs, rew, done, extras = env.step(pol)
    torch.cuda.synchronize()
    assert torch.equal(pol, keep)
    assert (task.actions[3] == 0).all() and float(rew[3]) == 0.0 and int(done[3]) == 1
    assert obs.shape == (n, 461) and float(obs.abs().max()) <= 5.0 and torch.isfinite(obs).all()
    assert rew.shape == (n,) and done.dtype == torch.int64
    assert set(extras) >= {"terminate", "sub_rewards", "sub_rewards_names"} and extras["sub_rewards"].shape == (n, 4)
    assert extras["sub_rewards_names"] == "dof_reward,vel_reward,body_pos_reward,body_rot_reward"
    assert int(task.progress_buf[0]) == 1
    # clip_actions: the task sees clamped actions
    env2 = VecTaskPythonWrapper(task, DEV, clip_observations=5.0, clip_actions=0.25)
    env2.step(pol)
    assert float(task.actions.abs().max()) <= 0.25
    task.close()


def test_reset_with_empty_and_partial_ids(mlib):
    n = 24
    task = make_task(n, mlib)
    task.reset()
    a = torch.cat([task._target_dof_pos.clone(), torch.zeros((n, 6), device=DEV)], dim=1).contiguous()
    for _ in range(3):
        task.step(a.clone())
    torch.cuda.synchronize()
    before = {k: N(getattr(task, k)).copy() for k in ("progress_buf", "reset_buf", "obs_buf", "_cur_ref_motion_times")}
    task.reset(torch.zeros(0, dtype=torch.long, device=DEV))  # nothing to do (humanoid_smpl.py:136-140)
    torch.cuda.synchronize()
    for k, v in before.items():
        assert np.array_equal(N(getattr(task, k)), v), k
    ids = torch.tensor([1, 7, 23], dtype=torch.long, device=DEV)
    task.reset(ids)
    torch.cuda.synchronize()
    prog = N(task.progress_buf)
    assert (prog[[1, 7, 23]] == 0).all() and (np.delete(prog, [1, 7, 23]) == 3).all()
    other = np.setdiff1d(np.arange(n), [1, 7, 23])
    assert np.array_equal(N(task.obs_buf)[other], before["obs_buf"][other])
    task.step(a.clone())
    torch.cuda.synchronize()
    assert torch.isfinite(task.obs_buf).all() and (N(task.progress_buf)[[1, 7, 23]] == 1).all()
    task.close()


def test_bad_action_tensors_are_refused(mlib):
    task = make_task(8, mlib)
    task.reset()
    with pytest.raises(RuntimeError):
        task.step(torch.zeros((8, 74), device=DEV))
    with pytest.raises(RuntimeError):
        task.step(torch.zeros((8, 75), device=DEV, dtype=torch.float64))
    with pytest.raises(RuntimeError):
        task.step(torch.zeros((8, 75)))  # host tensor
    with pytest.raises(RuntimeError):
        task.step(torch.zeros((8, 150), device=DEV)[:, ::2])  # not contiguous
    task.close()
