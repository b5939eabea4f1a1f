"""The ball's aerodynamic force in the C oracle against vectors recorded from the reference's own
HumanoidSMPLIMMVAE.apply_external_force_to_ball (tests/golden/ball_aero.npz, oracle/gen_golden_ball.py).  CPU only; the HIP kernel and the
bounce flags are checked against the same vectors in tests/test_gpu_racket_ball.py."""
import os

import numpy as np
import pytest

from oracle import phys_oracle as po

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ball_aero.npz"))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_aero_force_matches_reference(tag):
    st, want, scale = G["state_" + tag], G["force_" + tag], float(G["spin_scale_" + tag])
    got = np.stack([po.ball_aero(s, scale) for s in st])
    tol = 3e-6 * np.abs(want).max() + 1e-9  # the reference computes in float32
    assert np.abs(got - want).max() < tol, (np.abs(got - want).max(), tol)
    assert np.abs(want[0:4]).max() == 0.0 and np.abs(got[0:4]).max() == 0.0  # balls at rest: no force (the divide-by-zero guard)
    assert np.abs(want).max() > 0.1


@pytest.mark.parametrize("tag", ["a", "b"])
def test_golden_bounce_rule(tag):
    """What the recorded flags say (the rule the engine keeps inside its launch): a ball at or below 4 R (6 R with more than 2 substeps)
    that has not bounced yet is flagged, its position recorded."""
    st, had = G["state_" + tag], G["has_bounce_in_" + tag]
    thr = 0.032 * (6 if int(G["substeps_" + tag]) > 2 else 4)
    now = ~had & (st[:, 2] <= np.float32(thr))
    assert np.array_equal(G["has_bounce_now_" + tag], now) and np.array_equal(G["has_bounce_" + tag], had | now)
    assert np.array_equal(G["bounce_pos_" + tag][now], st[now, 0:3]) and np.all(G["bounce_pos_" + tag][~now] == 0)
    assert now.sum() > 5 and (~now).sum() > 5
