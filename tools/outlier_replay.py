#!/usr/bin/env python
"""Offline look at the envs tools/parity_sweep.py --dump found over the per-element bounds: the float64 oracle is re-run from the dumped
inputs (with the kernel's contact vertices), as it is and with float32 experiments switched on (oracle/phys v2p_oracle_experiment), and the
kernel's distance to each variant is printed.  usage: python tools/outlier_replay.py gpurun_out/outliers.npz [experiment flags, default 1]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.phys_oracle import BatchOracle, default_params, lib  # noqa: E402
from vid2player3d_amd.model import load_baked_model  # noqa: E402
from vid2player3d_amd.racket import with_racket  # noqa: E402

d = np.load(sys.argv[1])
flags = [int(x) for x in sys.argv[2:]] or [1]
names = sorted({k.split("/")[0] for k in d.files})
bm0 = load_baked_model()
bm_r, _ = with_racket(load_baked_model())
for name in names:
    g = lambda k: d["%s/%s" % (name, k)]
    n = len(g("env"))
    if n == 0:
        continue
    par = default_params(enable_contact=name != "pd only")
    par.solver_type = 1 if name == "tgs" else 0
    par.joint_limits = int(name.startswith("limits"))
    if name == "limits*":
        par.limit_margin = 1e9
    bm = bm_r if name.startswith("limits") else bm0
    res = {}
    for fl in [0] + flags:
        lib().v2p_oracle_experiment(C.c_int(fl))
        o = BatchOracle(bm, n, par)
        o.set_state(g("got/in_root"), g("got/in_dpos"), g("got/in_dvel"))
        res[fl] = o.step(g("got/pd"), g("got/force"), g("got/torque"), nsub=4, hold=2, forced_ids=g("got/ids_sub"))
    lib().v2p_oracle_experiment(C.c_int(0))
    k = g("got/dvel").astype(np.float64)
    assert np.abs(res[0]["dvel"] - g("ref/dvel")).max() < 1e-9, "replay does not reproduce the dumped oracle result"
    e0 = np.abs(k - res[0]["dvel"]).max(axis=1)
    line = "[replay] %-9s %3d envs: |kernel - oracle| dof vel: median %.1e max %.1e" % (name, n, np.median(e0), e0.max())
    for fl in flags:
        ef = np.abs(k - res[fl]["dvel"]).max(axis=1)
        eo = np.abs(res[fl]["dvel"] - res[0]["dvel"]).max(axis=1)
        line += " | exp %d: |kernel - oracle'| median %.1e max %.1e, |oracle' - oracle| median %.1e max %.1e, envs closer to the kernel by > 2x: %d" % (
            fl, np.median(ef), ef.max(), np.median(eo), eo.max(), int((ef < 0.5 * e0).sum()))
    print(line)
