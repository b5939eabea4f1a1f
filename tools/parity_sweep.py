#!/usr/bin/env python
"""Parity at scale: the HIP physics kernel against the float64 oracle (stepped with the kernel's contact vertices) on N perturbed states
per fixture, one control step each, with the per-element bounds of tests/test_gpu_physics.py - percentiles of every comparison and the
number of envs over the bounds.  usage (GPU box): python tools/parity_sweep.py [envs per fixture, default 2048]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_physics as T  # noqa: E402
from gpu_util import DEV, rows_close, synth_tables  # noqa: E402
from vid2player3d_amd.model import load_baked_model  # noqa: E402
from vid2player3d_amd.motion_lib import MotionLib  # noqa: E402
from vid2player3d_amd.racket import with_racket  # noqa: E402

mlib = MotionLib(synth_tables(seed=5, num_clips=8, min_frames=60, max_frames=120), DEV)
NENV = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bm_racket, _ = with_racket(load_baked_model())
FIX = (("pd only", dict(contact=False, seed=11, lift=0.3)), ("standing", dict(contact=True, seed=11, lift=0.0, vel_sigma=0.5)),
       ("fallen", dict(contact=True, seed=3, lift=-0.75, vel_sigma=0.2)), ("fast", dict(contact=True, seed=13, lift=-0.5, vel_sigma=3.0)),
       ("low", dict(contact=True, seed=17, lift=-0.9, vel_sigma=1.0)), ("tgs", dict(contact=True, seed=2, lift=-0.1, solver="tgs")),
       ("limits", dict(contact=True, seed=61, lift=0.0, vel_sigma=0.5, limits=True, body_model=bm_racket, act_sigma=0.5)),
       ("limits*", dict(contact=True, seed=61, lift=0.0, vel_sigma=0.5, limits=True, body_model=bm_racket, act_sigma=0.5, limit_margin=1e9)))  # rows always on
for name, kw in FIX:
    (got, ref), = T._run_pair(mlib, NENV, what=name, **kw)
    contact = kw["contact"]
    bad = rows_close(got["dvel"], ref["dvel"], T.VEL_ATOL, T.VEL_RTOL, name + " dof_vel")
    bad |= rows_close(got["rb"][..., 7:], ref["rb"][..., 7:], T.VEL_ATOL, T.VEL_RTOL, name + " rb vel")
    bad |= rows_close(got["df"], ref["df"], T.FORCE_ATOL, T.FORCE_RTOL, name + " dof force")
    if contact:
        bad |= rows_close(got["cf"], ref["cf"], T.FORCE_ATOL, T.FORCE_RTOL, name + " contact force")
    pos = np.abs(got["rb"][..., :3] - ref["rb"][..., :3]).max()
    coarse = np.abs(got["dvel"] - ref["dvel"]).max() / max(1.0, np.abs(ref["dvel"]).max())
    touched = (got["ids"] >= 0).any(axis=2).sum(axis=1)
    worst = np.argsort(-np.abs(got["dvel"] - ref["dvel"]).max(axis=1))[:3]
    print("[worst] %s: envs %s |ddvel| %s at dofs %s, clamp margins %s" % (name, worst.tolist(), np.abs(got["dvel"] - ref["dvel"]).max(axis=1)[worst], np.abs(got["dvel"] - ref["dvel"])[worst].argmax(axis=1), np.asarray(ref["clamp"])[worst]))
    print("[sweep] %-9s %d envs, touched links mean %.1f max %d: envs over the per-element bounds %d (%.2f %%); max |dpos| %.1e m; max |dvel| / max|vel| %.1e (coarse bound %.0e)"
          % (name, NENV, touched.mean(), touched.max(), int(bad.sum()), 100.0 * bad.mean(), pos, coarse, T.TOL_VEL))
