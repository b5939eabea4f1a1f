#!/usr/bin/env python
"""The conditioning of the model's control step as a statement about the MODEL (CPU only, the float64 oracle against itself).

The parity tests widen their per-element bounds by the oracle's own sensitivity (tests/gpu_util.py) because in ~1 % of perturbed
states one control step of this model amplifies a float32-rounding perturbation by 10^2 .. 10^4.  This tool answers what that is a
property OF:

  (a) the one-step gain (largest change of a joint rate / size of the velocity perturbation, 8 random directions) per fixture under
      PGS with 4 sweeps (the model), PGS with 8 and 16 sweeps, TGS with 4 slices, and PGS with friction off;
  (b) a converged reference (PGS, 200 sweeps per substep) and the distance of the 4-sweep result from it, per fixture, in the envs
      where 200 and 400 sweeps agree (box friction bounded by the current normal impulse does not converge everywhere).

usage: python tools/conditioning_report.py [envs per fixture, default 512]  ->  stdout (profiles/r05_conditioning.txt)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.phys_oracle import BatchOracle, default_params  # noqa: E402
from tools.gain_probe import fixture  # noqa: E402

FIX = (("standing", dict(seed=11, lift=0.0, vel_sigma=0.5)), ("fallen", dict(seed=3, lift=-0.75, vel_sigma=0.2)),
       ("fast", dict(seed=13, lift=-0.5, vel_sigma=3.0)), ("low", dict(seed=17, lift=-0.9, vel_sigma=1.0)))
VARIANTS = (("PGS 4 sweeps (the model)", dict()), ("PGS 8 sweeps", dict(n_iter=8)), ("PGS 16 sweeps", dict(n_iter=16)), ("PGS 64 sweeps", dict(n_iter=64)),
            ("TGS 4 slices", dict(solver_type=1)), ("PGS 4, friction off", dict(mu=0.0)), ("contacts off", dict(enable_contact=0)))


def gains(bm, n, state, par):
    root, dpos, dvel, pd, force, torque = state
    o = BatchOracle(bm, n, par, fast=True)
    o.set_state(root, dpos, dvel)
    s = o.sensitivity(pd, force, torque, trials=8, seed=1)
    return s["dvel"].max(axis=1) / 1e-6


def main(n):
    print("# conditioning of one control step (4 substeps of 1/120 s) of the engine's model, float64 oracle against itself, %d envs per fixture" % n)
    print("# gain = largest change of a joint rate over 8 perturbations of float32-rounding size (2e-7 on positions, 1e-6 on velocities) / 1e-6")
    print("\n## (a) one-step gain by solver variant: p50 / p90 / p99 / max over the envs")
    for name, kw in FIX:
        bm, *state = fixture(n, **kw)
        for label, pk in VARIANTS:
            g = gains(bm, n, state, default_params(**pk))
            print("%-9s %-26s p50 %8.1f  p90 %8.1f  p99 %9.0f  max %9.0f   envs with gain > 1000: %4d (%.2f %%)"
                  % (name, label, *np.percentile(g, [50, 90, 99, 100]), int((g > 1000).sum()), 100.0 * (g > 1000).mean()))
        print()
    print("## (b) distance of the model's 4 sweeps from the converged solution (PGS, 200 sweeps per substep), one SUBSTEP from the same state")
    print("#     (envs where 200 and 400 sweeps agree to 1e-6 rad/s; the others - box friction limits that move with the normal impulse - have no limit to compare with)")
    for name, kw in FIX:
        bm, root, dpos, dvel, pd, force, torque = fixture(n, **kw)

        def run(**pk):
            o = BatchOracle(bm, n, default_params(**pk), fast=True)
            o.set_state(root, dpos, dvel)
            return o.step(pd, force, torque, nsub=1, hold=1)
        c200, c400 = run(n_iter=200), run(n_iter=400)
        conv = np.abs(c200["dvel"] - c400["dvel"]).max(axis=1) < 1e-6
        touched = (c200["ids"] >= 0).any(axis=2).sum(axis=1)
        row = "%-9s touched links mean %.1f; converged envs %4d of %d |" % (name, touched.mean(), int(conv.sum()), n)
        for label, pk in (("PGS 4", dict()), ("PGS 8", dict(n_iter=8)), ("PGS 16", dict(n_iter=16)), ("PGS 64", dict(n_iter=64))):
            r = run(**pk)
            dv = np.abs(r["dvel"] - c200["dvel"]).max(axis=1)[conv]
            df = np.abs(r["cf"] - c200["cf"]).reshape(n, -1).max(axis=1)[conv]
            row += " %s: |d joint rate| p50 %.2e p99 %.2e rad/s, |d contact force| p50 %.1f p99 %.0f N |" % (label, *np.percentile(dv, [50, 99]), *np.percentile(df, [50, 99]))
        print(row)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 512)
