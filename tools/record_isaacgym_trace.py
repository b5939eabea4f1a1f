#!/usr/bin/env python
"""RUNS ON A MACHINE THAT HAS ISAAC GYM + THE REFERENCE (nv-tlabs/vid2player3d); it is not executed by this repository's tests.

Records a rollout of the reference's `HumanoidSMPLIM` into the .npz format of tests/golden/env_trace.npz, so that
`tools/replay_trace.py` can (a) teacher-force the recorded PhysX states through this engine's task ops (obs / reward / reset /
targets must match to float32 rounding) and (b) compare this engine's physics step against PhysX one control step at a time.
This is the procedure that PINS physics parity, which cannot be done inside the build container.

Usage, from `embodied_pose/` of the reference, after the task object exists (e.g. in a debugger or at the end of
`utils/parse_task.py:parse_task`):

    from record_isaacgym_trace import TraceRecorder
    rec = TraceRecorder(task)                 # task: env.tasks.humanoid_smpl_im.HumanoidSMPLIM
    ... run the usual loop (agent.play_steps or a manual `task.reset(); task.step(actions)` loop) for one or two epochs ...
    rec.save("trace.npz", motion_tables="mlib.npz")

The recorder only reads tensors; it changes nothing in the simulation.
"""
import numpy as np
import torch

STATE_NAMES = ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos", "rb_pos", "rb_rot")


def _np(t):
    return t.detach().cpu().numpy().copy()


class TraceRecorder:
    def __init__(self, task):
        self.task = task
        self.rec = {"motion_ids": _np(task._reset_ref_motion_ids)}
        self.epoch = -1
        self.step = 0
        self._orig_reset = task.reset
        self._orig_pre = task.pre_physics_step
        self._orig_post = task.post_physics_step
        task.reset = self._reset
        task.pre_physics_step = self._pre
        task.post_physics_step = self._post

    def _targets(self, prefix):
        t = self.task
        for n in STATE_NAMES:
            self.rec[prefix + "target_" + n] = _np(getattr(t, "_target_" + n))

    def _reset(self, env_ids=None):
        out = self._orig_reset(env_ids)
        t = self.task
        if env_ids is None or len(env_ids) == t.num_envs:  # the per-epoch reset of every env (im_agent.py:312)
            self.epoch += 1
            self.step = 0
            p = "e%d_reset_" % self.epoch
            n = t.num_envs
            self.rec["e%d_" % self.epoch + "reset_motion_times"] = _np(t._reset_ref_motion_times)
            self.rec[p + "root_states"] = _np(t._humanoid_root_states)
            self.rec[p + "dof_pos"] = _np(t._dof_pos)
            self.rec[p + "dof_vel"] = _np(t._dof_vel)
            self.rec[p + "rb_state"] = _np(t._rigid_body_state).reshape(n, -1, 13)
            self.rec["e%d_context_feat" % self.epoch] = _np(t.context_feat)
            self.rec["e%d_context_mask" % self.epoch] = _np(t.context_mask)
            self.rec[p + "obs"] = _np(t.obs_buf)
            for k, v in (("rew", t.rew_buf), ("reset", t.reset_buf), ("terminate", t._terminate_buf), ("progress", t.progress_buf),
                         ("cur_time", t._cur_ref_motion_times)):
                self.rec[p + k] = _np(v)
            self._targets(p)
        return out

    def _pre(self, actions):
        p = "e%d_s%02d_" % (max(self.epoch, 0), self.step)
        self.rec[p + "actions"] = _np(actions)
        out = self._orig_pre(actions)
        self.rec[p + "actions_after"] = _np(actions)  # masked in place (humanoid_smpl_im.py:126)
        if hasattr(self.task, "pd_torque"):
            self.rec[p + "pd_torque"] = _np(self.task.pd_torque)
        return out

    def _post(self):
        out = self._orig_post()
        t = self.task
        n = t.num_envs
        p = "e%d_s%02d_" % (max(self.epoch, 0), self.step)
        self.rec[p + "sim_dof_pos"] = _np(t._dof_pos)
        self.rec[p + "sim_dof_vel"] = _np(t._dof_vel)
        self.rec[p + "sim_rb_state"] = _np(t._rigid_body_state).reshape(n, -1, 13)
        self.rec[p + "sim_contact_force"] = _np(t._contact_forces)
        self.rec[p + "sub_rewards"] = _np(t.extras["sub_rewards"])
        self.rec[p + "obs"] = _np(t.obs_buf)
        for k, v in (("rew", t.rew_buf), ("reset", t.reset_buf), ("terminate", t._terminate_buf), ("progress", t.progress_buf),
                     ("cur_time", t._cur_ref_motion_times)):
            self.rec[p + k] = _np(v)
        self._targets(p)
        self.step += 1
        self.rec["num_steps" if self.epoch <= 0 else "num_steps_e%d" % self.epoch] = np.int64(self.step)
        return out

    def save(self, path, motion_tables=None):
        np.savez_compressed(path, **self.rec)
        if motion_tables:  # the flat motion tables this engine loads
            lib = self.task._motion_lib
            np.savez(motion_tables, **{k: _np(getattr(lib, k)) for k in ("gts", "grs", "lrs", "grvs", "gravs", "dvs")},
                     motion_lengths=_np(lib._motion_lengths), motion_num_frames=_np(lib._motion_num_frames), motion_dt=_np(lib._motion_dt),
                     motion_fps=_np(lib._motion_fps), motion_weights=_np(lib._motion_weights), motion_bodies=_np(lib._motion_bodies),
                     motion_min_verts_h=_np(lib._motion_min_verts_h))
