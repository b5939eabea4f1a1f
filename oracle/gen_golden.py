#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON (test infrastructure only).

Runs only where /root/reference exists (this build container).  The reference pins no
numerical results for the hot path, so the goldens are produced by importing
its code unmodified (through oracle/ref_shim) and recording inputs + outputs with fixed seeds:

  motion_tables.npz   synthetic clips pushed through the reference constructor path
                      SkeletonTree.from_mjcf -> SkeletonState -> SkeletonMotion -> MotionLib(dict)
                      (uhc/utils/convert_amass_isaac.py:134-176) -> gts/grs/lrs/grvs/gravs/dvs
  motion_state.npz    MotionLib.get_motion_state (embodied_pose/utils/motion_lib.py:164-266)
  task_ops.npz        compute_humanoid_reward / compute_humanoid_reset / dof_to_obs /
                      compute_humanoid_observations_imitation / pre-physics math
                      (embodied_pose/env/tasks/humanoid_smpl_im.py:125-157, 918-987)
  env_trace.npz       the reference HumanoidSMPLIM methods (reset / pre_physics_step /
                      post_physics_step) driven end to end on a gym-less instance whose
                      physics step is teacher-forced with recorded states

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

from ref_shim.install import install  # noqa: E402

install()

import torch  # noqa: E402

torch.set_num_threads(1)

from poselib.skeleton.skeleton3d import SkeletonTree, SkeletonState, SkeletonMotion  # noqa: E402
from utils.motion_lib import MotionLib  # noqa: E402
from utils import torch_utils as ref_tu  # noqa: E402
import env.tasks.humanoid_smpl_im as him  # noqa: E402
import env.tasks.humanoid_smpl as hs  # noqa: E402
from env.tasks.base_task import BaseTask  # noqa: E402

from vid2player3d_amd import synth  # noqa: E402
from vid2player3d_amd.model import load_baked_model  # noqa: E402

MJCF = "/root/reference/embodied_pose/data/assets/mjcf/smpl_mesh_humanoid_amass_v1.xml"
OUT = os.path.join(REPO, "tests", "golden")
KEY_BODY_IDS = [7, 3, 18, 23]  # R_Ankle, L_Ankle, L_Hand, R_Hand (amass_im.yaml:17)
DOF_BODY_IDS = list(range(1, 24))
DOF_OFFSETS = list(range(0, 70, 3))


def build_reference_motion_lib(clips):
    tree = SkeletonTree.from_mjcf(MJCF)
    d = {}
    for i, c in enumerate(clips):
        state = SkeletonState.from_rotation_and_root_translation(
            tree, torch.from_numpy(c["local_rot"]), torch.from_numpy(c["root_trans"]), is_local=True)
        motion = SkeletonMotion.from_skeleton_state(state, fps=c["fps"])
        out = motion.to_dict()
        out.update(seq_name="synth_%d" % i, seq_idx=i, pose_aa=np.zeros((c["local_rot"].shape[0], 72)),
                   beta=c["beta"], beta_idx=i, gender=c["gender"], min_verts_h=c["min_verts_h"],
                   body_scale=1.0, __name__="SkeletonMotion")
        d["synth_%d" % i] = out
    return MotionLib(motion_file=d, dof_body_ids=DOF_BODY_IDS, dof_offsets=DOF_OFFSETS,
                     key_body_ids=KEY_BODY_IDS, device="cpu", clean_up=True)


def npf(x):
    return x.detach().cpu().numpy()


def gen_motion(clips, mlib):
    tabs = {k: npf(getattr(mlib, k)) for k in ("gts", "grs", "lrs", "grvs", "gravs", "dvs")}
    tabs.update(
        motion_lengths=npf(mlib._motion_lengths), motion_num_frames=npf(mlib._motion_num_frames),
        motion_dt=npf(mlib._motion_dt), motion_fps=npf(mlib._motion_fps), motion_weights=npf(mlib._motion_weights),
        motion_bodies=npf(mlib._motion_bodies), motion_min_verts_h=npf(mlib._motion_min_verts_h),
        length_starts=npf(mlib.length_starts))
    for i, c in enumerate(clips):
        tabs["clip%d_local_rot" % i] = c["local_rot"]
        tabs["clip%d_root_trans" % i] = c["root_trans"]
        tabs["clip%d_beta" % i] = c["beta"]
        tabs["clip%d_min_verts_h" % i] = np.float64(c["min_verts_h"])
    np.savez_compressed(os.path.join(OUT, "motion_tables.npz"), **tabs)

    rng = np.random.default_rng(11)
    nclip = len(clips)
    lens = npf(mlib._motion_lengths)
    ids, times = [], []
    for c in range(nclip):  # edge cases: before start, exact frames, last frame, past the end
        for t in (-0.2, 0.0, 1.0 / 30, 0.5 / 30, lens[c] - 1.0 / 30, lens[c], lens[c] + 0.05, lens[c] + 0.4):
            ids.append(c)
            times.append(t)
    q_rand = 96
    rid = rng.integers(0, nclip, size=q_rand)
    ids += list(rid)
    times += list(rng.uniform(-0.1, 1.15, size=q_rand) * lens[rid])
    ids = torch.tensor(np.array(ids), dtype=torch.long)
    times = torch.tensor(np.array(times), dtype=torch.float32)
    names = ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos", "rb_pos", "rb_rot")
    out = {"ids": npf(ids), "times": npf(times), "ground_tolerance": np.float32(0.0)}
    res = mlib.get_motion_state(ids, times, return_rigid_body=True, adjust_height=True, ground_tolerance=0.0)
    out.update({n: npf(r) for n, r in zip(names, res)})
    res = mlib.get_motion_state(ids, times, return_rigid_body=True, adjust_height=False)
    out.update({n + "_noadj": npf(r) for n, r in zip(names, res)})
    np.savez_compressed(os.path.join(OUT, "motion_state.npz"), **out)


def rand_quat(rng, *shape):
    q = rng.normal(size=shape + (4,))
    return (q / np.linalg.norm(q, axis=-1, keepdims=True)).astype(np.float32)


def gen_task_ops():
    rng = np.random.default_rng(5)
    n = 48
    f32 = np.float32
    body_pos = rng.normal(0, 0.5, size=(n, 24, 3)).astype(f32)
    body_pos[..., 2] += 1.0
    body_rot = rand_quat(rng, n, 24)
    tgt_pos = (body_pos + rng.normal(0, 0.05, size=body_pos.shape)).astype(f32)
    small = rng.normal(0, 0.08, size=(n, 24, 3))
    dq = synth._quat_from_rotvec(small)
    tgt_rot = synth._quat_mul(dq, body_rot.astype(np.float64)).astype(f32)
    tgt_rot[0] = body_rot[0]            # identical rotations: exercises the acos(1) / tiny-sin branch
    tgt_rot[1, :12] = -body_rot[1, :12]  # antipodal representation of the same rotation
    dof_pos = rng.normal(0, 0.6, size=(n, 69)).astype(f32)
    dof_pos[2, :9] = 0.0                # zero exp-map: default-axis branch of exp_map_to_angle_axis
    dof_vel = rng.normal(0, 2.0, size=(n, 69)).astype(f32)
    tgt_dof_pos = (dof_pos + rng.normal(0, 0.1, size=dof_pos.shape)).astype(f32)
    tgt_dof_vel = (dof_vel + rng.normal(0, 0.5, size=dof_vel.shape)).astype(f32)
    body_vel = rng.normal(0, 1.0, size=(n, 24, 3)).astype(f32)
    body_ang_vel = rng.normal(0, 2.0, size=(n, 24, 3)).astype(f32)
    weights = np.ones(24, dtype=f32)
    weights[[13, 18, 23]] = [2.0, 1.5, 1.5]
    specs = {'k_dof': 60.0, 'k_vel': 0.2, 'k_pos': 100.0, 'k_rot': 40.0, 'w_dof': 0.6, 'w_vel': 0.1, 'w_pos': 0.2, 'w_rot': 0.1}
    T = torch.from_numpy
    out = dict(body_pos=body_pos, body_rot=body_rot, tgt_pos=tgt_pos, tgt_rot=tgt_rot, dof_pos=dof_pos, dof_vel=dof_vel,
               tgt_dof_pos=tgt_dof_pos, tgt_dof_vel=tgt_dof_vel, body_vel=body_vel, body_ang_vel=body_ang_vel,
               body_pos_weights=weights)
    rew, sub, names = him.compute_humanoid_reward(T(body_pos), T(body_rot), T(tgt_pos), T(tgt_rot), T(dof_pos), T(dof_vel),
                                                  T(tgt_dof_pos), T(tgt_dof_vel), T(body_vel), T(body_ang_vel), 138,
                                                  DOF_OFFSETS, T(weights), specs)
    out.update(reward=npf(rew), sub_rewards=npf(sub), sub_rewards_names=np.array(names))
    out["dof_obs"] = npf(hs.dof_to_obs(T(dof_pos), 138, DOF_OFFSETS))

    # compute_humanoid_reset (humanoid_smpl_im.py:956-987)
    reset_in = torch.zeros(n, dtype=torch.long)
    progress = torch.tensor(rng.integers(0, 6, size=n), dtype=torch.long)
    progress[5] = 299
    progress[6] = 298
    heights = np.full(24, -0.5, dtype=f32)
    heights[13] = 1.0
    rb = body_pos.copy()
    rb[:, 13, 2] = rng.uniform(0.8, 1.6, size=n)   # head height straddles the 1.0 m threshold
    rb[7, 7, 2] = -0.7                              # contact body below its threshold: must be ignored
    rb[8, 2, 2] = -0.6                              # non-contact body below -0.5: terminates
    cur_t = rng.uniform(0.0, 3.0, size=n).astype(f32)
    clip_len = rng.uniform(1.0, 4.0, size=n).astype(f32)
    cur_t[9] = clip_len[9]                          # equality counts as "reached"
    rst, term = him.compute_humanoid_reset(reset_in, progress, torch.zeros(n, 24, 3), torch.tensor([7, 3]), T(rb), 300.0, True,
                                           T(heights), T(cur_t), T(clip_len))
    out.update(reset_progress=npf(progress), reset_rb_pos=rb, reset_heights=heights, reset_cur_time=cur_t,
               reset_clip_len=clip_len, reset_out=npf(rst), terminate_out=npf(term))

    # pre-physics math, composed exactly as humanoid_smpl_im.py:125-157 composes the reference helpers
    actions = rng.normal(0, 1.0, size=(n, 75)).astype(f32)
    actions[:, :69] = dof_pos + rng.normal(0, 1.2, size=(n, 69))  # some beyond +-pi/2 of q -> clamp
    reset_mask = np.zeros(n, dtype=np.int64)
    reset_mask[[3, 11]] = 1
    a = T(actions.copy())
    a[T(reset_mask) == 1] = 0
    pd_lim = 0.5 * np.pi
    pd_tar = torch.maximum(torch.minimum(a[:, :69], T(dof_pos) + pd_lim), T(dof_pos) - pd_lim)
    kp = torch.linspace(100.0, 1100.0, 69)
    pd_torque = (pd_tar - T(dof_pos)) * kp
    res_f = a[:, 69:72].clone() * 31.85
    res_t = a[:, 72:75].clone() * 31.85
    root_rot = him.remove_base_rot(T(body_rot[:, 0, :]))
    hq = ref_tu.calc_heading_quat(root_rot)
    out.update(pre_actions=actions, pre_reset=reset_mask, pre_actions_masked=npf(a), pre_pd_tar=npf(pd_tar), pre_kp=npf(kp),
               pre_pd_torque=npf(pd_torque), pre_res_force=npf(ref_tu.my_quat_rotate(hq, res_f)),
               pre_res_torque=npf(ref_tu.my_quat_rotate(hq, res_t)), pre_heading_quat=npf(hq))

    # 734-d in-network observation (next row f-1; humanoid_smpl_im.py:773-850, same code as
    # embodied_pose/models/im_network_builder.py:262-338)
    mb = rng.normal(size=(n, 11)).astype(f32)
    o734 = him.compute_humanoid_observations_imitation(T(body_pos), T(body_rot), T(tgt_pos), T(tgt_rot), T(dof_pos), T(dof_vel), T(tgt_dof_pos), T(body_vel),
                      T(body_ang_vel), T(mb), True, True)
    out.update(obs734_motion_bodies=mb, obs734=npf(o734))
    # RunningNorm in eval mode on top of it (embodied_pose/models/running_norm.py:32-43)
    from models.running_norm import RunningNorm
    rn = RunningNorm(734)
    rn.train()
    rn(o734 * 1.0)
    rn(o734 * 0.5 + 0.1)
    rn.eval()
    out.update(rn_mean=npf(rn.mean), rn_std=npf(rn.std), obs734_normed=npf(rn(o734)))

    # GAE scan (embodied_pose/learning/common_agent.py:423-435), called unbound on a stand-in for the agent
    from learning.common_agent import CommonAgent
    tt, ne = 32, 40
    fd = (rng.uniform(size=(tt, ne)) < 0.1).astype(f32)
    vals = rng.normal(size=(tt, ne, 1)).astype(f32)
    rews = rng.uniform(0, 1, size=(tt, ne, 1)).astype(f32)
    nvals = rng.normal(size=(tt, ne, 1)).astype(f32)
    agent = types.SimpleNamespace(horizon_length=tt, gamma=0.99, tau=0.95)
    advs = CommonAgent.discount_values(agent, T(fd), T(vals), T(rews), T(nvals))
    out.update(gae_fdones=fd, gae_values=vals, gae_rewards=rews, gae_next_values=nvals, gae_advs=npf(advs), gae_gamma=np.float32(0.99), gae_tau=np.float32(0.95))
    np.savez_compressed(os.path.join(OUT, "task_ops.npz"), **out)


class _GymSink:
    """Accepts every gym.* call of the per-step / per-reset paths and does nothing."""

    def __getattr__(self, name):
        return lambda *a, **k: None


def make_gymless_task(mlib, n, motion_ids, body_model):
    """A reference HumanoidSMPLIM instance built without Isaac Gym: every attribute that
    __init__/_create_envs/_setup_tensors would have produced is set by hand from the same
    sources (amass_im.yaml, the baked body model)."""
    task = object.__new__(him.HumanoidSMPLIM)
    args = types.SimpleNamespace(test=False)
    task.cfg = {"env": {"numEnvs": n}, "args": args}
    task.args = args
    task.device = "cpu"
    task.model = None
    task.gym = _GymSink()
    task.sim = None
    task.viewer = None
    task.debug_viz = False
    task.num_envs = n
    task.num_bodies = 24
    task.num_dof = 69
    task._num_dof = 69
    task._pd_control = True
    task.power_scale = 1.0
    task.residual_force_scale = 31.85
    task.residual_torque_scale = 31.85
    task.context_length = 32
    task.context_padding = 8
    task.truncate_time = True
    task.pd_tar_lim = 0.5 * np.pi
    task.control_freq_inv = 2
    task.dt = 2 * (1.0 / 60.0)
    task._motion_sync_dt = task.dt
    task._state_init = him.HumanoidSMPLIM.StateInit.Hybrid
    task._hybrid_init_prob = 1.0
    task.ground_tolerance = 0.0
    task._motion_lib = mlib
    task.max_episode_length = 300
    task._enable_early_termination = True
    task._local_root_obs = True
    task._root_height_obs = True
    task.dr_randomizations = {}
    task.extras = {}
    task.body_names = body_model.body_names
    task._dof_body_ids = DOF_BODY_IDS
    task._dof_offsets = DOF_OFFSETS
    task._dof_obs_size = 138
    task.obs_names = ['body_pos', 'body_rot', 'dof_pos', 'dof_vel', 'body_vel', 'body_ang_vel', 'motion_bodies']
    task.context_names = ['body_pos', 'body_rot', 'dof_pos', 'body_pos_gt', 'dof_pos_gt']
    task.is_env_dim_setup = False
    task.stiffness = torch.from_numpy(body_model.kp.astype(np.float32))
    task.damping = torch.from_numpy(body_model.kd.astype(np.float32))
    task.body_pos_weights = torch.ones(24)
    th = np.full(24, -0.5)
    th[13] = max(1.0, th[13])
    task._termination_heights = torch.tensor(th, dtype=torch.float32)
    task._contact_body_ids = torch.tensor([7, 3], dtype=torch.long)
    task._key_body_ids = torch.tensor(KEY_BODY_IDS, dtype=torch.long)
    # sim state tensors (humanoid_smpl.py:66-113)
    task._root_states = torch.zeros(n, 13)
    task._humanoid_root_states = task._root_states
    task._humanoid_actor_ids = torch.arange(n, dtype=torch.int32)
    task._dof_state = torch.zeros(n * 69, 2)
    task._dof_pos = task._dof_state.view(n, 69, 2)[..., 0]
    task._dof_vel = task._dof_state.view(n, 69, 2)[..., 1]
    task._rigid_body_state = torch.zeros(n * 24, 13)
    rbs = task._rigid_body_state.view(n, 24, 13)
    task._rigid_body_pos = rbs[..., 0:3]
    task._rigid_body_rot = rbs[..., 3:7]
    task._rigid_body_vel = rbs[..., 7:10]
    task._rigid_body_ang_vel = rbs[..., 10:13]
    task._contact_forces = torch.zeros(n, 24, 3)
    task.dof_force_tensor = torch.zeros(n, 69)
    # buffers (base_task.py:62-74, humanoid_smpl.py:52)
    task.obs_buf = torch.zeros(n, 461)
    task.states_buf = torch.zeros(n, 0)
    task.rew_buf = torch.zeros(n)
    task.reset_buf = torch.ones(n, dtype=torch.long)
    task.progress_buf = torch.zeros(n, dtype=torch.long)
    task._terminate_buf = torch.ones(n, dtype=torch.long)
    task._reset_ref_motion_ids = torch.tensor(motion_ids, dtype=torch.long)
    task._reset_ref_motion_bodies = mlib._motion_bodies[task._reset_ref_motion_ids]
    task._reset_default_env_ids = []
    task._reset_ref_env_ids = []
    task._sub_rewards = None
    task._sub_rewards_names = None
    task._state_reset_happened = False
    return task


def gen_env_trace(clips, mlib, body_model):
    torch.manual_seed(7)
    rng = np.random.default_rng(23)
    n, steps = 6, 36
    motion_ids = np.array([0, 1, 2, 0, 1, 2])
    task = make_gymless_task(mlib, n, motion_ids, body_model)

    forced = {}

    def teacher_forced_physics(self):
        s = forced["cur"]
        self._dof_pos[:] = torch.from_numpy(s["dof_pos"])
        self._dof_vel[:] = torch.from_numpy(s["dof_vel"])
        self._rigid_body_state.view(n, 24, 13)[:] = torch.from_numpy(s["rb_state"])
        self._humanoid_root_states[:] = torch.from_numpy(s["rb_state"][:, 0, :])

    task._physics_step = types.MethodType(teacher_forced_physics, task)

    rec = {"motion_ids": motion_ids, "num_steps": np.int64(steps)}

    def snapshot(prefix):
        rec[prefix + "obs"] = npf(task.obs_buf).copy()
        rec[prefix + "rew"] = npf(task.rew_buf).copy()
        rec[prefix + "reset"] = npf(task.reset_buf).copy()
        rec[prefix + "terminate"] = npf(task._terminate_buf).copy()
        rec[prefix + "progress"] = npf(task.progress_buf).copy()
        rec[prefix + "cur_time"] = npf(task._cur_ref_motion_times).copy()
        for k in ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos", "rb_pos", "rb_rot"):
            rec[prefix + "target_" + k] = npf(getattr(task, "_target_" + k)).copy()

    def run_epoch(tag, first_step, nsteps):
        task.reset()  # HumanoidSMPL.reset(None): all envs (humanoid_smpl.py:136-140)
        rec[tag + "reset_motion_times"] = npf(task._reset_ref_motion_times).copy()
        rec[tag + "reset_root_states"] = npf(task._humanoid_root_states).copy()
        rec[tag + "reset_dof_pos"] = npf(task._dof_pos).copy()
        rec[tag + "reset_dof_vel"] = npf(task._dof_vel).copy()
        rec[tag + "reset_rb_state"] = npf(task._rigid_body_state.view(n, 24, 13)).copy()
        rec[tag + "context_feat"] = npf(task.context_feat).copy()
        rec[tag + "context_mask"] = npf(task.context_mask).copy()
        snapshot(tag + "reset_")
        for i in range(nsteps):
            t = first_step + i
            # the "simulated" state: the current target perturbed, as a tracking controller would leave it
            tgt_pos = npf(task._target_rb_pos).astype(np.float64)
            tgt_rot = npf(task._target_rb_rot).astype(np.float64)
            rb = np.zeros((n, 24, 13), dtype=np.float32)
            rb[..., 0:3] = tgt_pos + rng.normal(0, 0.03, size=tgt_pos.shape)
            dq = synth._quat_from_rotvec(rng.normal(0, 0.06, size=(n, 24, 3)))
            rb[..., 3:7] = synth._quat_mul(dq, tgt_rot)
            rb[..., 7:10] = rng.normal(0, 0.5, size=(n, 24, 3))
            rb[..., 10:13] = rng.normal(0, 1.0, size=(n, 24, 3))
            if t >= 9:
                rb[2, 13, 2] = 0.6   # env 2 drops its head below 1.0 m -> early termination
            if t in (20, 21):
                rb[4, 13, 2] = 0.9   # env 4 dips for two steps: sticky reset must hold afterwards
            s = {
                "dof_pos": (npf(task._target_dof_pos) + rng.normal(0, 0.05, size=(n, 69))).astype(np.float32),
                "dof_vel": (npf(task._target_dof_vel) + rng.normal(0, 0.3, size=(n, 69))).astype(np.float32),
                "rb_state": rb,
            }
            forced["cur"] = s
            actions = (npf(task._target_dof_pos)[:, :69] + rng.normal(0, 0.17, size=(n, 69))).astype(np.float32)
            actions = np.concatenate([actions, rng.normal(0, 0.17, size=(n, 6)).astype(np.float32)], axis=1)
            a = torch.from_numpy(actions.copy())
            BaseTask.step(task, a)
            p = "%ss%02d_" % (tag, i)
            rec[p + "actions"] = actions
            rec[p + "actions_after"] = npf(a).copy()        # masked in place (humanoid_smpl_im.py:126)
            rec[p + "pd_torque"] = npf(task.pd_torque).copy()
            rec[p + "sim_dof_pos"] = s["dof_pos"]
            rec[p + "sim_dof_vel"] = s["dof_vel"]
            rec[p + "sim_rb_state"] = s["rb_state"]
            rec[p + "sub_rewards"] = npf(task._sub_rewards).copy()
            snapshot(p)

    run_epoch("e0_", 0, steps)
    run_epoch("e1_", 0, 4)   # second epoch: reset of all envs clears the sticky flags
    rec["num_steps_e1"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, "env_trace.npz"), **rec)


def main():
    os.makedirs(OUT, exist_ok=True)
    body_model = load_baked_model()
    clips = synth.make_clips(seed=3, num_clips=3, min_frames=34, max_frames=60)
    mlib = build_reference_motion_lib(clips)
    gen_motion(clips, mlib)
    gen_task_ops()
    gen_env_trace(clips, mlib, body_model)
    for f in sorted(os.listdir(OUT)):
        print("%-24s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024.0))


if __name__ == "__main__":
    main()
