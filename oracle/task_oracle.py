"""CPU restatement (numpy, float32) of the NON-physics half of the hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker.  The product path (vid2player3d_amd/) never imports this.

Pinned: every function here is checked against golden vectors recorded from the reference's
own Python (tests/golden/*.npz, made by oracle/gen_golden.py) in tests/test_oracle_golden.py.

Each function cites the reference lines it restates (paths relative to /root/reference).
Quaternions are xyzw.
"""
import numpy as np

F = np.float32
PI = np.float32(np.pi)


# --------------------------------------------------------------------------------------------
# quaternion helpers  (embodied_pose/utils/torch_utils.py, isaacgym.torch_utils)
# --------------------------------------------------------------------------------------------
def quat_mul(a, b):
    """Hamilton product, xyzw (isaacgym.torch_utils.quat_mul; same value as
    poselib/poselib/core/rotation3d.py:15-27)."""
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
        w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2,
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
    ], axis=-1).astype(F)


def quat_conjugate(q):
    return np.concatenate([-q[..., :3], q[..., 3:]], axis=-1)


def normalize_angle(x):
    """isaacgym.torch_utils.normalize_angle = atan2(sin x, cos x)."""
    return np.arctan2(np.sin(x), np.cos(x)).astype(F)


def quat_from_angle_axis(angle, axis):
    """isaacgym.torch_utils.quat_from_angle_axis (axis normalised, result re-normalised)."""
    theta = (angle / F(2))[..., None]
    axis = axis / np.maximum(np.linalg.norm(axis, axis=-1, keepdims=True), F(1e-9))
    q = np.concatenate([axis * np.sin(theta), np.cos(theta)], axis=-1).astype(F)
    return (q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), F(1e-9))).astype(F)


def my_quat_rotate(q, v):
    """embodied_pose/utils/torch_utils.py:70-79."""
    qw = q[..., 3:4]
    qv = q[..., :3]
    a = v * (F(2) * qw * qw - F(1))
    b = np.cross(qv, v) * qw * F(2)
    c = qv * np.sum(qv * v, axis=-1, keepdims=True) * F(2)
    return (a + b + c).astype(F)


def quat_to_angle_axis(q):
    """embodied_pose/utils/torch_utils.py:82-102."""
    qw = q[..., 3]
    sin_theta = np.sqrt(np.maximum(F(1) - qw * qw, F(0)).astype(F))  # torch sqrt of a tiny negative gives nan only if qw>1; goldens stay in range
    sin_theta = np.where(F(1) - qw * qw < 0, np.float32(np.nan), sin_theta)
    with np.errstate(divide="ignore", invalid="ignore"):
        angle = normalize_angle(F(2) * np.arccos(qw))  # |qw|>1 -> nan, masked to 0 below exactly as torch does
        axis = q[..., :3] / sin_theta[..., None]
    mask = np.abs(sin_theta) > F(1e-5)
    default_axis = np.zeros_like(axis)
    default_axis[..., 2] = 1
    angle = np.where(mask, angle, F(0))
    axis = np.where(mask[..., None], axis, default_axis)
    return angle.astype(F), axis.astype(F)


def quat_to_exp_map(q):
    """embodied_pose/utils/torch_utils.py:113-119."""
    angle, axis = quat_to_angle_axis(q)
    return (angle[..., None] * axis).astype(F)


def quat_to_tan_norm(q):
    """embodied_pose/utils/torch_utils.py:122-134: rotated x axis then rotated z axis."""
    tan = np.zeros(q.shape[:-1] + (3,), dtype=F)
    tan[..., 0] = 1
    nrm = np.zeros_like(tan)
    nrm[..., 2] = 1
    return np.concatenate([my_quat_rotate(q, tan), my_quat_rotate(q, nrm)], axis=-1)


def exp_map_to_angle_axis(e):
    """embodied_pose/utils/torch_utils.py:144-160."""
    angle = np.linalg.norm(e, axis=-1).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        axis = e / angle[..., None]
    angle = normalize_angle(angle)
    default_axis = np.zeros_like(e)
    default_axis[..., 2] = 1
    mask = np.abs(angle) > F(1e-5)
    angle = np.where(mask, angle, F(0))
    axis = np.where(mask[..., None], axis, default_axis)
    return angle.astype(F), axis.astype(F)


def exp_map_to_quat(e):
    """embodied_pose/utils/torch_utils.py:163-166."""
    angle, axis = exp_map_to_angle_axis(e)
    return quat_from_angle_axis(angle, axis)


def slerp(q0, q1, t):
    """embodied_pose/utils/torch_utils.py:169-190 (t broadcast over the last axis)."""
    cos_half = np.sum(q0 * q1, axis=-1, keepdims=True)
    q1 = np.where(cos_half < 0, -q1, q1)
    cos_half = np.abs(cos_half)
    half = np.arccos(np.minimum(cos_half, F(1)))  # torch.acos(>1) = nan, overwritten by the >=1 branch below
    sin_half = np.sqrt(np.maximum(F(1) - cos_half * cos_half, F(0)))
    with np.errstate(divide="ignore", invalid="ignore"):
        ra = np.sin((F(1) - t) * half) / sin_half
        rb = np.sin(t * half) / sin_half
        new_q = ra * q0 + rb * q1
    new_q = np.where(np.abs(sin_half) < F(0.001), F(0.5) * q0 + F(0.5) * q1, new_q)
    new_q = np.where(np.abs(cos_half) >= 1, q0, new_q)
    return new_q.astype(F)


def calc_heading(q):
    """embodied_pose/utils/torch_utils.py:193-204."""
    ref = np.zeros(q.shape[:-1] + (3,), dtype=F)
    ref[..., 0] = 1
    rot = my_quat_rotate(q, ref)
    return np.arctan2(rot[..., 1], rot[..., 0]).astype(F)


def calc_heading_quat(q):
    """embodied_pose/utils/torch_utils.py:206-217."""
    axis = np.zeros(q.shape[:-1] + (3,), dtype=F)
    axis[..., 2] = 1
    return quat_from_angle_axis(calc_heading(q), axis)


def calc_heading_quat_inv(q):
    """embodied_pose/utils/torch_utils.py:219-243."""
    axis = np.zeros(q.shape[:-1] + (3,), dtype=F)
    axis[..., 2] = 1
    h = calc_heading(q)
    return quat_from_angle_axis(-h, axis), h


def remove_base_rot(q):
    """embodied_pose/env/tasks/humanoid_smpl_im.py:766-770."""
    base = quat_conjugate(np.array([0.5, 0.5, 0.5, 0.5], dtype=F))
    return quat_mul(q, np.broadcast_to(base, q.shape))


# --------------------------------------------------------------------------------------------
# reference-motion sampler  (embodied_pose/utils/motion_lib.py)
# --------------------------------------------------------------------------------------------
KEY_BODY_IDS = (7, 3, 18, 23)


def calc_frame_blend(time, length, num_frames, dt):
    """motion_lib.py:427-436.  blend is NOT clamped (it exceeds 1 past the clip end)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        phase = np.clip((time / length).astype(F), F(0), F(1))
    f0 = (phase * (num_frames - 1).astype(F)).astype(np.int64)
    f1 = np.minimum(f0 + 1, num_frames - 1)
    blend = ((time - f0.astype(F) * dt) / dt).astype(F)
    return f0, f1, blend


def local_rotation_to_dof(local_rot):
    """motion_lib.py:460-488 for 23 three-dof joints: exp-map of bodies 1..23."""
    n = local_rot.shape[0]
    return quat_to_exp_map(local_rot[:, 1:, :]).reshape(n, -1)


def get_motion_state(tabs, ids, times, adjust_height=True, ground_tolerance=0.0, key_body_ids=KEY_BODY_IDS):
    """motion_lib.py:164-266 with return_rigid_body=True.  `tabs` = dict of the flat tables."""
    ids = np.asarray(ids, dtype=np.int64)
    times = np.asarray(times, dtype=F)
    length = tabs["motion_lengths"][ids]
    nf = tabs["motion_num_frames"][ids]
    dt = tabs["motion_dt"][ids]
    f0, f1, blend = calc_frame_blend(times, length, nf, dt)
    f0l = f0 + tabs["length_starts"][ids]
    f1l = f1 + tabs["length_starts"][ids]
    gts, grs, lrs = tabs["gts"], tabs["grs"], tabs["lrs"]
    b1 = blend[:, None]
    b2 = blend[:, None, None]
    root_pos = ((F(1) - b1) * gts[f0l, 0] + b1 * gts[f1l, 0]).astype(F)
    root_rot = slerp(grs[f0l, 0], grs[f1l, 0], b1)
    kb = list(key_body_ids)
    key_pos = ((F(1) - b2) * gts[f0l][:, kb] + b2 * gts[f1l][:, kb]).astype(F)
    local_rot = slerp(lrs[f0l], lrs[f1l], b2)
    dof_pos = local_rotation_to_dof(local_rot)
    root_vel = tabs["grvs"][f0l]          # velocities come from frame 0 only (no blend)
    root_ang_vel = tabs["gravs"][f0l]
    dof_vel = tabs["dvs"][f0l]
    rb_pos = ((F(1) - b2) * gts[f0l] + b2 * gts[f1l]).astype(F)
    rb_rot = slerp(grs[f0l], grs[f1l], b2)
    if adjust_height:
        min_vh = (tabs["motion_min_verts_h"][ids] - F(ground_tolerance)).astype(F)
        root_pos[..., 2] -= min_vh
        key_pos[..., 2] -= min_vh[:, None]
        rb_pos[..., 2] -= min_vh[:, None]
    return root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos, rb_pos, rb_rot


MOTION_STATE_NAMES = ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos", "rb_pos", "rb_rot")


# --------------------------------------------------------------------------------------------
# task ops  (embodied_pose/env/tasks/humanoid_smpl_im.py, humanoid_smpl.py)
# --------------------------------------------------------------------------------------------
def dof_to_obs(pose):
    """humanoid_smpl.py:604-635 for 23 three-dof joints -> [N,138]."""
    n = pose.shape[0]
    q = exp_map_to_quat(pose.reshape(n, -1, 3))
    return quat_to_tan_norm(q).reshape(n, -1)


REWARD_SPECS = {"k_dof": 60.0, "k_vel": 0.2, "k_pos": 100.0, "k_rot": 40.0, "w_dof": 0.6, "w_vel": 0.1, "w_pos": 0.2, "w_rot": 0.1}


def compute_humanoid_reward(body_pos, body_rot, tgt_pos, tgt_rot, dof_pos, dof_vel, tgt_dof_pos, tgt_dof_vel, body_pos_weights, specs=None):
    """humanoid_smpl_im.py:918-953."""
    s = dict(REWARD_SPECS)
    s.update(specs or {})
    diff = dof_to_obs(dof_pos) - dof_to_obs(tgt_dof_pos)
    dof_r = np.exp(-F(s["k_dof"]) * np.mean(diff * diff, axis=-1, dtype=F))
    dv = tgt_dof_vel - dof_vel
    vel_r = np.exp(-F(s["k_vel"]) * np.mean(dv * dv, axis=-1, dtype=F))
    dp = (tgt_pos - body_pos) * body_pos_weights[:, None]
    pos_r = np.exp(-F(s["k_pos"]) * np.mean(np.mean(dp * dp, axis=-1, dtype=F), axis=-1, dtype=F))
    dq = quat_mul(tgt_rot, quat_conjugate(body_rot))
    ang = quat_to_angle_axis(dq)[0]
    rot_r = np.exp(-F(s["k_rot"]) * np.mean(ang * ang, axis=-1, dtype=F))
    rew = F(s["w_dof"]) * dof_r + F(s["w_vel"]) * vel_r + F(s["w_pos"]) * pos_r + F(s["w_rot"]) * rot_r
    return rew.astype(F), np.stack([dof_r, vel_r, pos_r, rot_r], axis=-1).astype(F)


def compute_humanoid_reset(progress, rb_pos, term_heights, cur_time, clip_len, contact_body_ids=(7, 3), max_episode_length=300.0,
                           enable_early_termination=True):
    """humanoid_smpl_im.py:956-987 (the contact-force term is commented out in the reference)."""
    n = progress.shape[0]
    terminated = np.zeros(n, dtype=np.int64)
    if enable_early_termination:
        fall = rb_pos[..., 2] < term_heights
        fall[:, list(contact_body_ids)] = False
        fallen = np.any(fall, axis=-1) & (progress > 1)
        terminated = np.where(fallen, 1, terminated)
    reset_cond = (progress >= max_episode_length - 1) | (cur_time >= clip_len)
    return np.where(reset_cond, 1, terminated).astype(np.int64), terminated.astype(np.int64)


def pre_physics(actions, reset_buf, dof_pos, root_body_rot, kp, pd_tar_lim=0.5 * np.pi, res_force_scale=31.85, res_torque_scale=31.85):
    """humanoid_smpl_im.py:125-157, 391-396.  Returns (masked actions, pd target, pd torque, world force, world torque)."""
    a = actions.astype(F).copy()
    a[reset_buf == 1] = 0
    lim = F(pd_tar_lim)
    pd_tar = np.maximum(np.minimum(a[:, :69], dof_pos + lim), dof_pos - lim).astype(F)
    pd_torque = ((pd_tar - dof_pos) * kp).astype(F)
    hq = calc_heading_quat(remove_base_rot(root_body_rot))
    force = my_quat_rotate(hq, a[:, 69:72] * F(res_force_scale))
    torque = my_quat_rotate(hq, a[:, 72:75] * F(res_torque_scale))
    return a, pd_tar, pd_torque, force, torque


def humanoid_obs(rb_pos, rb_rot, dof_pos, dof_vel, rb_vel, rb_ang_vel, motion_bodies):
    """humanoid_smpl_im.py:653-668 with obs_names of :198 -> [N,461]."""
    n = rb_pos.shape[0]
    return np.concatenate([rb_pos.reshape(n, -1), rb_rot.reshape(n, -1), dof_pos, dof_vel, rb_vel.reshape(n, -1),
                           rb_ang_vel.reshape(n, -1), motion_bodies], axis=-1).astype(F)


def init_context(tabs, motion_ids, motion_times, dt, context_length=32, context_padding=8, ground_tolerance=0.0):
    """humanoid_smpl_im.py:530-563 -> context_feat[N,48,378], context_mask[N,48]."""
    n = motion_ids.shape[0]
    padded = context_length + 2 * context_padding
    t0 = (motion_times + F(dt)).astype(F)
    steps = (F(dt) * np.arange(-context_padding, context_length + context_padding).astype(F)).astype(F)
    all_t = (t0[:, None] + steps[None, :]).astype(F)
    all_ids = np.repeat(motion_ids[:, None], padded, axis=1)
    res = get_motion_state(tabs, all_ids.reshape(-1), all_t.reshape(-1), True, ground_tolerance)
    dof_pos, rb_pos, rb_rot = res[2], res[7], res[8]
    q = n * padded
    feat = np.concatenate([rb_pos.reshape(q, -1), rb_rot.reshape(q, -1), dof_pos, rb_pos.reshape(q, -1), dof_pos], axis=-1)
    mask = all_t <= (tabs["motion_lengths"][motion_ids] + F(2) * F(dt))[:, None]
    return feat.reshape(n, padded, -1).astype(F), mask


def heading_to_vec(h):
    """embodied_pose/utils/torch_transform.py heading_to_vec: [cos h, sin h]."""
    return np.stack([np.cos(h), np.sin(h)], axis=-1).astype(F)


def obs_imitation_734(body_pos, body_rot, tgt_pos, tgt_rot, dof_pos, dof_vel, tgt_dof_pos, body_vel, body_ang_vel, motion_bodies):
    """humanoid_smpl_im.py:773-850 (= embodied_pose/models/im_network_builder.py:262-338) with
    local_root_obs=True, root_height_obs=True.  Reproduces the reference's `root_rot_obs`
    overwrite (:807-808): tan-norm of the un-headed root rotation."""
    n, b = body_pos.shape[:2]
    root_pos = body_pos[:, 0]
    root_rot = remove_base_rot(body_rot[:, 0])
    root_h = root_pos[:, 2:3]
    hinv, heading = calc_heading_quat_inv(root_rot)
    hexp = np.repeat(hinv[:, None, :], b, axis=1)
    local_pos = my_quat_rotate(hexp, body_pos - root_pos[:, None]).reshape(n, -1)[:, 3:]
    local_rot_obs = quat_to_tan_norm(quat_mul(hexp, body_rot)).reshape(n, -1)
    local_rot_obs[:, 0:6] = quat_to_tan_norm(root_rot)
    local_vel = my_quat_rotate(hexp, body_vel).reshape(n, -1)
    local_ang = my_quat_rotate(hexp, body_ang_vel).reshape(n, -1)
    t_root_pos = tgt_pos[:, 0]
    t_root_rot = remove_base_rot(tgt_rot[:, 0])
    rel_h = root_h - t_root_pos[:, 2:3]
    _, t_heading = calc_heading_quat_inv(t_root_rot)
    rel_root_rot = quat_to_tan_norm(quat_mul(t_root_rot, quat_conjugate(root_rot)))
    rel_2d = my_quat_rotate(hinv, t_root_pos - root_pos)[:, :2]
    rel_head = heading_to_vec(t_heading - heading)
    rel_dof = tgt_dof_pos - dof_pos
    rel_body_pos = my_quat_rotate(hexp, tgt_pos - body_pos).reshape(n, -1)
    rel_body_rot = quat_to_tan_norm(quat_mul(quat_conjugate(body_rot), tgt_rot)).reshape(n, -1)
    return np.concatenate([root_h, local_pos, local_rot_obs, local_vel, local_ang, dof_vel, rel_h, rel_root_rot, rel_2d, rel_head,
                           rel_dof, rel_body_pos, rel_body_rot, motion_bodies], axis=-1).astype(F)


def running_norm_eval(x, mean, std, clip=5.0):
    """embodied_pose/models/running_norm.py:32-43 in eval mode with n > 0."""
    return np.clip((x - mean) / (std + F(1e-8)), -F(clip), F(clip)).astype(F)


def discount_values(fdones, values, rewards, next_values, gamma, tau):
    """GAE reverse scan, embodied_pose/learning/common_agent.py:423-435.  [T,N], [T,N,1] x3 -> [T,N,1]."""
    t_len = rewards.shape[0]
    advs = np.zeros_like(rewards)
    last = np.zeros_like(rewards[0])
    for t in reversed(range(t_len)):
        not_done = (F(1) - fdones[t])[:, None]
        delta = rewards[t] + F(gamma) * next_values[t] - values[t]
        last = (delta + F(gamma) * F(tau) * not_done * last).astype(F)
        advs[t] = last
    return advs


# --------------------------------------------------------------------------------------------
# the task state machine around the physics step
# --------------------------------------------------------------------------------------------
class TaskOracle:
    """HumanoidSMPLIM's reset / pre_physics_step / post_physics_step bookkeeping
    (humanoid_smpl_im.py:125-157, 398-418, 442-468, 489-528, 594-624, 670-692, 724-755;
    humanoid_smpl.py:153-173) with the physics step supplied by the caller: either recorded
    states (teacher forcing, tests/golden/env_trace.npz) or oracle/phys (the C restatement)."""

    def __init__(self, tabs, motion_ids, kp, body_pos_weights=None, term_heights=None, dt=1.0 / 30.0, ground_tolerance=0.0,
                 max_episode_length=300.0, context_length=32, context_padding=8):
        self.tabs = tabs
        self.n = n = len(motion_ids)
        self.motion_ids = np.asarray(motion_ids, dtype=np.int64)
        self.kp = np.asarray(kp, dtype=F)
        self.dt = F(dt)
        self.ground_tolerance = ground_tolerance
        self.max_episode_length = max_episode_length
        self.context_length = context_length
        self.context_padding = context_padding
        self.body_pos_weights = np.ones(24, dtype=F) if body_pos_weights is None else np.asarray(body_pos_weights, dtype=F)
        if term_heights is None:
            term_heights = np.full(24, -0.5, dtype=F)
            term_heights[13] = 1.0
        self.term_heights = np.asarray(term_heights, dtype=F)
        self.motion_bodies = tabs["motion_bodies"][self.motion_ids]
        self.root_states = np.zeros((n, 13), dtype=F)
        self.dof_pos = np.zeros((n, 69), dtype=F)
        self.dof_vel = np.zeros((n, 69), dtype=F)
        self.rb_state = np.zeros((n, 24, 13), dtype=F)
        self.obs_buf = np.zeros((n, 461), dtype=F)
        self.rew_buf = np.zeros(n, dtype=F)
        self.sub_rewards = np.zeros((n, 4), dtype=F)
        self.reset_buf = np.ones(n, dtype=np.int64)
        self.terminate_buf = np.ones(n, dtype=np.int64)
        self.progress_buf = np.zeros(n, dtype=np.int64)
        self.cur_time = np.zeros(n, dtype=F)
        self.target = None
        self.prev_target = None
        self._state_reset_happened = False

    def _set_target(self):
        self.target = get_motion_state(self.tabs, self.motion_ids, (self.cur_time + self.dt).astype(F), True, self.ground_tolerance)

    def reset_all(self, motion_times):
        """_reset_envs for all envs with explicit RSI times (humanoid_smpl_im.py:442-450, 489-528)."""
        mt = np.asarray(motion_times, dtype=F)
        st = get_motion_state(self.tabs, self.motion_ids, mt, True, self.ground_tolerance)
        root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, _, rb_pos, rb_rot = st
        self.root_states[:, 0:3] = root_pos
        self.root_states[:, 3:7] = root_rot
        self.root_states[:, 7:10] = root_vel
        self.root_states[:, 10:13] = root_ang_vel
        self.rb_state[..., 0:3] = rb_pos
        self.rb_state[..., 3:7] = rb_rot
        self.rb_state[..., 7:13] = 0
        self.dof_pos[:] = dof_pos
        self.dof_vel[:] = dof_vel
        self.cur_time = mt.copy()
        self._set_target()
        self.context_feat, self.context_mask = init_context(self.tabs, self.motion_ids, mt, self.dt, self.context_length,
                                                            self.context_padding, self.ground_tolerance)
        self.progress_buf[:] = 0
        self.reset_buf[:] = 0
        self.terminate_buf[:] = 0
        self.obs_buf = humanoid_obs(self.rb_state[..., 0:3], self.rb_state[..., 3:7], self.dof_pos, self.dof_vel,
                                    self.rb_state[..., 7:10], self.rb_state[..., 10:13], self.motion_bodies)

    def pre_physics_step(self, actions):
        out = pre_physics(actions, self.reset_buf, self.dof_pos, self.rb_state[:, 0, 3:7], self.kp)
        self.actions, self.pd_tar, self.pd_torque, self.res_force, self.res_torque = out
        self.prev_target = tuple(x.copy() for x in self.target)
        return out

    def set_sim_state(self, dof_pos, dof_vel, rb_state):
        self.dof_pos[:] = dof_pos
        self.dof_vel[:] = dof_vel
        self.rb_state[:] = rb_state
        self.root_states[:] = rb_state[:, 0, :]

    def post_physics_step(self):
        self.progress_buf += 1
        self.cur_time = (self.cur_time + self.dt).astype(F)
        self._set_target()
        self.obs_buf = humanoid_obs(self.rb_state[..., 0:3], self.rb_state[..., 3:7], self.dof_pos, self.dof_vel,
                                    self.rb_state[..., 7:10], self.rb_state[..., 10:13], self.motion_bodies)
        pt = self.prev_target
        rew, sub = compute_humanoid_reward(self.rb_state[..., 0:3], self.rb_state[..., 3:7], pt[7], pt[8], self.dof_pos, self.dof_vel,
                                           pt[2], pt[5], self.body_pos_weights)
        mask = self.reset_buf == 1
        rew[mask] = 0
        sub[mask] = 0
        self.rew_buf, self.sub_rewards = rew, sub
        old_reset, old_term = self.reset_buf.copy(), self.terminate_buf.copy()
        rst, term = compute_humanoid_reset(self.progress_buf, self.rb_state[..., 0:3], self.term_heights, self.cur_time,
                                           self.tabs["motion_lengths"][self.motion_ids], max_episode_length=self.max_episode_length)
        rst[old_reset == 1] = 1
        term[old_reset == 1] = old_term[old_reset == 1]
        self.reset_buf, self.terminate_buf = rst, term
