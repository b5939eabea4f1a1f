#!/usr/bin/env python
"""The reference's OWN task code timed on host cores: the non-physics part of one env step of the imitation task -
`HumanoidSMPLIM.pre_physics_step` (a2), `post_physics_step` = `_refresh_sim_tensors` + next target from `MotionLib.get_motion_state`
(a5, a6) + `_compute_humanoid_obs` (a8) + `compute_humanoid_reward` (a9) + `compute_humanoid_reset` (a10), and the per-epoch `reset()`
(RSI, a11, + `_init_context`, a12) - imported unmodified from /root/reference (through oracle/ref_shim; the @torch.jit.script functions
compile as in the reference) and run on a gym-less task instance (the harness of oracle/gen_golden.py) at the bench's shape: 8192 envs,
64 synthetic clips, 32-step epochs.  The physics call between pre and post is a teacher-forced state copy (Isaac Gym is closed; the
reference's CPU pipeline would run PhysX there), so this is the reference's PyTorch-CPU cost of everything BUT the physics: an upper
bound on the env-steps/s its CPU pipeline can reach.

Runs where /root/reference exists (the build container; /root/reference does not travel to the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tools/ref_cpu_baseline.py [--threads 8] [--envs 8192] [--epochs 2]   -> profiles/r03_ref_cpu_task_ops.json
bench.py attaches the committed file to its line as cpu_baseline.reference_task_ops (labelled with host and core count)."""
import argparse
import json
import os
import platform
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.dont_write_bytecode = True

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r03_ref_cpu_task_ops.json"))
args = ap.parse_args()

import gen_golden as G  # noqa: E402  (installs the shim, imports the reference's modules)
import torch  # noqa: E402

from vid2player3d_amd import synth  # noqa: E402
from vid2player3d_amd.model import load_baked_model  # noqa: E402

torch.set_num_threads(args.threads)
n, H = args.envs, 32
bm = load_baked_model()
clips = synth.make_clips(7, 64, 90, 300)  # the bench's 64 seeded synthetic clips
mlib = G.build_reference_motion_lib(clips)
rng = np.random.default_rng(7)
task = G.make_gymless_task(mlib, n, rng.integers(0, 64, size=n), bm)
noise = [torch.from_numpy(rng.normal(0, 0.17, size=(n, 75)).astype(np.float32)) for _ in range(H)]


def teacher_forced_physics(self):
    # stands in for gym.simulate x2 + the six refresh calls: the state a tracking controller would leave (cheap copies, timed with the rest)
    self._dof_pos[:] = self._target_dof_pos
    self._dof_vel[:] = self._target_dof_vel
    rbs = self._rigid_body_state.view(n, 24, 13)
    rbs[..., 0:3] = self._target_rb_pos
    rbs[..., 3:7] = self._target_rb_rot
    self._humanoid_root_states[:] = rbs[:, 0, :]


task._physics_step = types.MethodType(teacher_forced_physics, task)
t_reset = t_step = 0.0
parts = {"pre_physics_step": 0.0, "post_physics_step": 0.0}
for ep in range(args.epochs + 1):  # epoch 0 = warm-up (jit compilation)
    t0 = time.perf_counter()
    task.reset()
    t1 = time.perf_counter()
    pre = post = 0.0
    for k in range(H):
        a = noise[k].clone()
        a[:, :69] += task._target_dof_pos
        s0 = time.perf_counter()
        task.pre_physics_step(a)
        s1 = time.perf_counter()
        task._physics_step()
        s2 = time.perf_counter()
        task.post_physics_step()
        s3 = time.perf_counter()
        pre += s1 - s0
        post += s3 - s2
    t2 = time.perf_counter()
    if ep > 0:
        t_reset += t1 - t0
        t_step += t2 - t1
        parts["pre_physics_step"] += pre
        parts["post_physics_step"] += post
frames = args.epochs * H * n
out = {"what": "the reference's own HumanoidSMPLIM.reset / pre_physics_step / post_physics_step (imported from /root/reference, unmodified) on CPU tensors; "
               "physics replaced by a teacher-forced state copy", "envs": n, "epochs_timed": args.epochs, "horizon": H,
       "env_steps_per_s_task_ops_only": frames / (t_reset + t_step), "ms_per_step": 1e3 * t_step / (args.epochs * H), "ms_per_epoch_reset": 1e3 * t_reset / args.epochs,
       "ms_pre_physics_step": 1e3 * parts["pre_physics_step"] / (args.epochs * H), "ms_post_physics_step": 1e3 * parts["post_physics_step"] / (args.epochs * H),
       "threads": args.threads, "host": platform.node(), "cpu": platform.processor() or platform.machine(), "torch": torch.__version__,
       "source": "tools/ref_cpu_baseline.py, run in the build container (the reference tree does not exist on the GPU box)"}
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(out, open(args.out, "w"), indent=1)
print(json.dumps(out))
