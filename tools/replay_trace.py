#!/usr/bin/env python
"""Replay a recorded rollout (tests/golden/env_trace.npz format; tools/record_isaacgym_trace.py writes it on an Isaac Gym box)
through this engine on an MI355X:

  (a) task ops, teacher-forced: the recorded simulated states go in through the state tensors, obs / reward / reset / targets that
      come out are compared with the recording (float32 rounding expected);
  (b) physics, one control step at a time: the recorded state of step i is pushed, the recorded actions applied, one physics step
      run, and the result compared with the recorded state of step i+1 (this is the number that pins - or quantifies the distance of -
      the engine's own physics model against PhysX; with the synthetic golden trace it only shows that the tool runs).

    python tools/replay_trace.py trace.npz --motion mlib.npz [--body-model blob.npz] [--friction-frame world|velocity] [--solver pgs|tgs]
"""
import argparse
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vid2player3d_amd.model import BodyModel, load_baked_model  # noqa: E402
from vid2player3d_amd.motion_lib import MotionLib  # noqa: E402
from vid2player3d_amd.tasks import HumanoidSMPLIM, default_cfg  # noqa: E402

NAMES = ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos", "rb_pos", "rb_rot")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--motion", required=True, help="flat motion tables (.npz) of the recorded run")
    ap.add_argument("--body-model", help="compiled body model blob (.npz); default: the baked amass_v1 asset")
    ap.add_argument("--friction-frame", choices=["world", "velocity"], default="world",
                    help="tangent frame of the hull x ground friction rows (v2p_sim_cfg.friction_frame): run both and keep the one the trace agrees with")
    ap.add_argument("--solver", choices=["pgs", "tgs"], default=None, help="contact solver (default: the engine's, PGS); the reference's yaml names TGS")
    args = ap.parse_args()
    dev = "cuda:0"
    T = lambda x, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(x)).to(device=dev, dtype=dt).contiguous()  # noqa: E731
    N = lambda t: t.detach().cpu().numpy()  # noqa: E731
    g = dict(np.load(args.trace))
    with np.load(args.motion) as z:
        lib = MotionLib({k: z[k] for k in z.files}, dev)
    n = len(g["motion_ids"])
    cfg = default_cfg(n, motion_lib=lib, record_pd_torque=True, motion_ids=g["motion_ids"], body_shape_mismatch="warn")  # a trace names its own body; the baked one is used here
    cfg["env"]["friction_frame"] = args.friction_frame
    if args.solver:
        cfg["env"]["contact_solver"] = args.solver
    if args.body_model:
        with np.load(args.body_model) as z:
            cfg["env"]["body_model"] = BodyModel({k: z[k] for k in z.files})
    task = HumanoidSMPLIM(cfg, device_type="cuda", device_id=0)
    worst = {}

    def dev_(name, got, ref):
        e = float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max()) if np.size(ref) else 0.0
        worst[name] = max(worst.get(name, 0.0), e)

    epoch = 0
    while "e%d_reset_motion_times" % epoch in g:
        tag = "e%d_" % epoch
        steps = int(g["num_steps" if epoch == 0 else "num_steps_e%d" % epoch])
        task.reset_with_times(None, T(g[tag + "reset_motion_times"]))
        dev_("reset obs", N(task.obs_buf), g[tag + "reset_obs"])
        prev = None
        for i in range(steps):
            p = "%ss%02d_" % (tag, i)
            a = T(g[p + "actions"])
            # (b) free-running physics from the recorded previous state
            if prev is not None:
                task._dof_pos[:] = T(g[prev + "sim_dof_pos"])
                task._dof_vel[:] = T(g[prev + "sim_dof_vel"])
                task._humanoid_root_states[:] = T(g[prev + "sim_rb_state"][:, 0, :])
                task._reset_env_tensors(None)
                task.pre_physics_step(a.clone())
                task._physics_step()
                rb = N(task._rigid_body_state).reshape(n, -1, 13)
                alive = g[prev + "reset"] == 0
                dev_("physics: body position after one step [m]", rb[alive][..., :3], g[p + "sim_rb_state"][alive][..., :3])
                dev_("physics: body linear velocity [m/s]", rb[alive][..., 7:10], g[p + "sim_rb_state"][alive][..., 7:10])
                dev_("physics: dof position [rad]", N(task._dof_pos)[alive], g[p + "sim_dof_pos"][alive])
            # (a) teacher-forced task ops
            task.pre_physics_step(a)
            dev_("actions masked in place", N(a), g[p + "actions_after"])
            task._dof_pos[:] = T(g[p + "sim_dof_pos"])
            task._dof_vel[:] = T(g[p + "sim_dof_vel"])
            task._rigid_body_state.view(n, -1, 13)[:] = T(g[p + "sim_rb_state"])
            task._humanoid_root_states[:] = T(g[p + "sim_rb_state"][:, 0, :])
            task._reset_env_tensors(None, with_rb_state=True)
            task.post_physics_step()
            dev_("obs", N(task.obs_buf), g[p + "obs"])
            dev_("reward", N(task.rew_buf), g[p + "rew"])
            dev_("reset flags", N(task.reset_buf), g[p + "reset"])
            dev_("terminate flags", N(task.extras["terminate"]), g[p + "terminate"])
            for name in NAMES:
                dev_("target " + name, N(getattr(task, "_target_" + name)), g[p + "target_" + name])
            prev = p
        epoch += 1
    torch.cuda.synchronize()
    print("max |engine - recording| over %d epoch(s), %d envs:" % (epoch, n))
    for k in sorted(worst):
        print("  %-48s %.3e" % (k, worst[k]))
    task.close()


if __name__ == "__main__":
    main()
