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


def test_default_pose_and_hybrid_state_init(mlib):
    """stateInit 'Default' and 'Hybrid' with hybridInitProb < 1 (humanoid_smpl_im.py:470-487, 638-651): a default-reset env gets the
    actor's start pose (root at 0.89 m, identity rotation, joints and velocities at zero), cleared flags, and an observation assembled
    from the tensors as they are - the rigid-body tensor keeps the bodies of before the reset until the next physics step, as in the
    reference; reference-state envs of a hybrid reset are untouched by it."""
    n = 64
    task = make_task(n, mlib, stateInit="Default")
    task.reset()
    torch.cuda.synchronize()
    root = N(task._humanoid_root_states)
    assert np.allclose(root[:, :3], [0.0, 0.0, 0.89]) and np.allclose(root[:, 3:7], [0, 0, 0, 1]) and np.all(root[:, 7:] == 0)
    assert np.all(N(task._dof_pos) == 0) and np.all(N(task._dof_vel) == 0)
    assert int(task.reset_buf.sum()) == 0 and int(task.progress_buf.sum()) == 0 and int(task._terminate_buf.sum()) == 0
    obs = N(task.obs_buf)
    rb = N(task._rigid_body_state).reshape(n, 24, 13)
    assert np.array_equal(obs[:, :72], rb[:, :, 0:3].reshape(n, -1)) and np.array_equal(obs[:, 72:168], rb[:, :, 3:7].reshape(n, -1))
    assert np.all(obs[:, 168:306] == 0) and np.array_equal(obs[:, 450:461], N(task._reset_ref_motion_bodies)[:, :11])
    # at creation the rigid-body tensor holds the start pose with all joints at zero: the pelvis at 0.89 m, unit quaternions
    assert np.allclose(rb[:, 0, :3], [0.0, 0.0, 0.89]) and np.allclose(rb[:, :, 3:7], [0, 0, 0, 1]) and rb[:, :, 2].min() > 0.0
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    for _ in range(3):
        task.step(torch.cat([0.1 * torch.randn((n, 69), device=DEV, generator=g), torch.zeros((n, 6), device=DEV)], dim=1).contiguous())
    torch.cuda.synchronize()
    assert torch.isfinite(task.obs_buf).all() and torch.isfinite(task._rigid_body_state).all()
    assert float(task._rigid_body_pos[:, 0, 2].min()) > 0.3, "the humanoids stand / settle from the start pose"
    task.close()

    task = make_task(n, mlib, stateInit="Hybrid", hybridInitProb=0.5)
    torch.manual_seed(11)
    task.reset()
    torch.cuda.synchronize()
    dflt = (N(task._dof_pos) == 0).all(axis=1) & np.isclose(N(task._humanoid_root_states)[:, 2], 0.89)
    assert 8 < dflt.sum() < n - 8, "both kinds of reset happen at hybridInitProb 0.5 (got %d default of %d)" % (dflt.sum(), n)
    ref = ~dflt
    assert (N(task._cur_ref_motion_times)[ref] > 0).all(), "reference-state envs start inside their clip"
    assert (np.abs(N(task._dof_pos)[ref]).max(axis=1) > 0).all()
    task.step(torch.zeros((n, 75), device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(task.obs_buf).all()
    task.close()


def test_smpl_back_channels_are_in_smpl_joint_order(mlib):
    """smpl_rest_joints / smpl_parents / smpl_children (humanoid_smpl_im.py:325-327 -> im_agent.py:100-102): SMPL joint order
    (smpl_parser.py:10-35), the SMPL kinematic tree, and the first-child map with its two exceptions (smpl_parser.py:340-350)."""
    from vid2player3d_amd.tasks.humanoid_smpl_im import SMPL_BONE_ORDER_NAMES

    task = make_task(4, mlib)
    assert task.smpl_parents.tolist() == [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
    ch = task.smpl_children.tolist()
    assert ch[0] == 3 and ch[9] == 12 and ch[1] == 4 and ch[12] == 15 and ch[13] == 16 and all(ch[i] == -1 for i in (10, 11, 15, 22, 23))
    rest = N(task.smpl_rest_joints)
    assert rest.shape == (4, 24, 3)
    bm = task.body_model
    names = list(bm.body_names)
    # joint i sits at its parent's position + the MJCF offset of the body with the same name
    for i, nm in enumerate(SMPL_BONE_ORDER_NAMES):
        b = names.index(nm)
        p = task.smpl_parents[i].item()
        want = np.asarray(bm.local_pos[b]) + (rest[0, p] if p >= 0 else 0.0)
        assert np.allclose(rest[0, i], want, atol=1e-6), nm
    lh, rh = rest[0, SMPL_BONE_ORDER_NAMES.index("L_Hip")] - rest[0, 0], rest[0, SMPL_BONE_ORDER_NAMES.index("R_Hip")] - rest[0, 0]
    k = int(np.argmax(np.abs(lh - rh)))  # the lateral axis: the hips sit on opposite sides of the pelvis
    assert abs(lh[k] - rh[k]) > 0.1 and lh[k] * rh[k] < 0
    task.close()
