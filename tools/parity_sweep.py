#!/usr/bin/env python
"""Parity at scale: the HIP physics kernel against the float64 oracle (stepped with the kernel's contact vertices) on N perturbed states
per fixture, one control step each, with the per-element bounds of tests/test_gpu_physics.py - percentiles of every comparison and the
number of envs over the bounds.  usage (GPU box): python tools/parity_sweep.py [envs per fixture, default 2048] [--schedule env_per_lane]
[--fixtures standing,fallen] [--dump file.npz: inputs, kernel and oracle outputs of every env over the bounds, for offline analysis with
the CPU oracle (tools/outlier_replay.py)]"""
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
import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("nenv", nargs="?", type=int, default=2048)
ap.add_argument("--schedule", default="link_per_lane")
ap.add_argument("--fixtures", default="")
ap.add_argument("--dump", default="")
args = ap.parse_args()
NENV = args.nenv
DUMP = {}
bm_racket, _ = with_racket(load_baked_model())
FIX = (("pd only", dict(contact=False, seed=11, lift=0.3)), ("standing", dict(contact=True, seed=11, lift=0.0, vel_sigma=0.5)),
       ("fallen", dict(contact=True, seed=3, lift=-0.75, vel_sigma=0.2)), ("fast", dict(contact=True, seed=13, lift=-0.5, vel_sigma=3.0)),
       ("low", dict(contact=True, seed=17, lift=-0.9, vel_sigma=1.0)), ("tgs", dict(contact=True, seed=2, lift=-0.1, solver="tgs")),
       ("limits", dict(contact=True, seed=61, lift=0.0, vel_sigma=0.5, limits=True, body_model=bm_racket, act_sigma=0.5)),
       ("limitstgs", dict(contact=True, seed=61, lift=0.0, vel_sigma=0.5, limits=True, solver="tgs", body_model=bm_racket, act_sigma=0.5)),
       ("limits*", dict(contact=True, seed=61, lift=0.0, vel_sigma=0.5, limits=True, body_model=bm_racket, act_sigma=0.5, limit_margin=1e9)))  # rows always on
for name, kw in FIX:
    if args.fixtures and name not in args.fixtures.split(","):
        continue
    if args.schedule != "link_per_lane":
        kw = dict(kw, kernel_schedule=args.schedule)
    (got, ref), = T._run_pair(mlib, NENV, what=name, **kw)
    contact = kw["contact"]
    # flat per-element bounds (what rounds 2-3 counted) ...
    flat = rows_close(got["dvel"], ref["dvel"], T.VEL_ATOL, T.VEL_RTOL, name + " dof_vel (flat)")
    flat |= rows_close(got["rb"][..., 7:], ref["rb"][..., 7:], T.VEL_ATOL, T.VEL_RTOL, name + " rb vel (flat)")
    flat |= rows_close(got["df"], ref["df"], T.FORCE_ATOL, T.FORCE_RTOL, name + " dof force (flat)")
    if contact:
        flat |= rows_close(got["cf"], ref["cf"], T.FORCE_ATOL, T.FORCE_RTOL, name + " contact force (flat)")
    # ... and the conditioning-aware ones the tests assert on every env (tests/test_gpu_physics.py: K_SENS x the oracle's own sensitivity)
    bad = T.rows_all(got, ref, name, contact)
    S = ref["sens"]
    e_v = np.abs(got["dvel"] - ref["dvel"])
    ratio = (e_v / np.maximum(S["dvel"], 1e-12))[flat] if flat.any() else np.zeros(1)
    amp = S["dvel"].max(axis=1) / 1e-6   # gain of the oracle's own step on a 1e-6 perturbation, per env
    print("[cond] %-9s oracle's own gain (max |d dof_vel| / 1e-6 input perturbation) per env: p50 %.1f p99 %.1f max %.0f; among the %d envs over the flat bounds: "
          "median gain %.0f, kernel error / sensitivity at their dof_vel entries p50 %.2f max %.1f"
          % (name, np.percentile(amp, 50), np.percentile(amp, 99), amp.max(), int(flat.sum()), np.median(amp[flat]) if flat.any() else 0.0, np.percentile(ratio, 50), ratio.max()))
    pos = np.abs(got["rb"][..., :3] - ref["rb"][..., :3]).max()
    coarse = np.abs(got["dvel"] - ref["dvel"]).max() / max(1.0, np.abs(ref["dvel"]).max())
    touched = (got["ids"] >= 0).any(axis=2).sum(axis=1)
    if args.dump:
        sel = np.nonzero(flat)[0]
        for k, v in got.items():
            DUMP["%s/got/%s" % (name, k)] = np.asarray(v)[sel]
        for k in ("root", "dpos", "dvel", "rb", "cf", "df", "clamp"):
            DUMP["%s/ref/%s" % (name, k)] = np.asarray(ref[k])[sel]
        DUMP["%s/env" % name] = sel
    worst = np.argsort(-np.abs(got["dvel"] - ref["dvel"]).max(axis=1))[:3]
    print("[worst] %s: envs %s |ddvel| %s at dofs %s, clamp margins %s" % (name, worst.tolist(), np.abs(got["dvel"] - ref["dvel"]).max(axis=1)[worst], np.abs(got["dvel"] - ref["dvel"])[worst].argmax(axis=1), np.asarray(ref["clamp"])[worst]))
    print("[sweep] %-9s %d envs, touched links mean %.1f max %d: envs over the FLAT per-element bounds %d (%.2f %%), over the conditioning-aware bounds (asserted by the tests) %d; max |dpos| %.1e m; max |dvel| / max|vel| %.1e"
          % (name, NENV, touched.mean(), touched.max(), int(flat.sum()), 100.0 * flat.mean(), int(bad.sum()), pos, coarse))
if args.dump:
    np.savez_compressed(args.dump, **DUMP)
