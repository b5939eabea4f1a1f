"""Golden vectors for the ball's aerodynamic force and the bounce flags, recorded by RUNNING THE REFERENCE'S OWN METHOD
HumanoidSMPLIMMVAE.apply_external_force_to_ball (/root/reference/vid2player/env/tasks/humanoid_smpl_im_mvae.py:711-739, with get_cd / get_cl /
kf / R of vid2player/utils/tennis_ball.py) on a gym-less stand-in for `self`.  TEST INFRASTRUCTURE; runs in the build container only:
    PYTORCH_JIT=0 PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ball.py        -> tests/golden/ball_aero.npz
(PYTORCH_JIT=0: the module's @torch.jit.script helpers are not needed and some do not compile under this torch.)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ.setdefault("PYTORCH_JIT", "0")

from ref_shim import install as shim  # noqa: E402

shim._STUB_ROOTS = shim._STUB_ROOTS + ("smpl_visualizer",)  # third-party viewer the module imports at the top (absent here, never called)
shim.install()
sys.path.insert(0, "/root/reference/vid2player")

import torch  # noqa: E402
import env.tasks.humanoid_smpl_im_mvae as ref  # noqa: E402

torch.manual_seed(11)
out = {}
n = 96
for tag, substeps, spin_scale in (("a", 2, 1.0), ("b", 4, 0.6)):
    st = torch.zeros((n, 13))
    st[:, 0:3] = torch.randn(n, 3) * 3
    st[:, 2] = torch.rand(n) * 0.4  # around the bounce thresholds 4 R = 0.128 / 6 R = 0.192
    st[:, 6] = 1
    st[:, 7:10] = torch.randn(n, 3) * torch.tensor([15.0, 15.0, 6.0])
    st[:, 10:13] = torch.randn(n, 3) * 120
    st[0:4, 7:10] = 0            # balls at rest (the divide-by-zero guard)
    st[4:8, 10:13] = 0           # no spin
    st[8, 7:10] = torch.tensor([0.0, 0.0, -20.0])   # straight down: velocity parallel to g
    has = torch.rand(n) < 0.3
    me = types.SimpleNamespace(num_envs=n, device="cpu", cfg_v2p={"spin_scale": spin_scale}, cfg={"sim": {"substeps": substeps}},
                               _has_bounce=has.clone(), _has_bounce_now=torch.zeros(n, dtype=torch.bool), _bounce_pos=torch.zeros((n, 3)),
                               forces=torch.zeros((n, 26, 3)))
    ref.HumanoidSMPLIMMVAE.apply_external_force_to_ball(me, st)
    out["state_" + tag] = st.numpy().astype(np.float32)
    out["has_bounce_in_" + tag] = has.numpy()
    out["force_" + tag] = me.forces[:, -1].numpy().astype(np.float32)
    out["has_bounce_" + tag] = me._has_bounce.numpy()
    out["has_bounce_now_" + tag] = me._has_bounce_now.numpy()
    out["bounce_pos_" + tag] = me._bounce_pos.numpy().astype(np.float32)
    out["substeps_" + tag] = np.int64(substeps)
    out["spin_scale_" + tag] = np.float64(spin_scale)
    assert float(me.forces[:, :-1].abs().max()) == 0.0
np.savez(os.path.join(HERE, "..", "tests", "golden", "ball_aero.npz"), **out)
print("wrote tests/golden/ball_aero.npz", {k: getattr(v, "shape", v) for k, v in out.items()})
