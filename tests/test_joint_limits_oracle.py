"""Joint-limit rows of the physics oracle (oracle/phys/v2p_phys_oracle.c, v2p_oparams.joint_limits): invariants on the CPU.
The HIP kernel is compared with this oracle in tests/test_gpu_physics.py::test_joint_limits_match_oracle."""
import numpy as np
import pytest

from oracle import phys_oracle as po
from vid2player3d_amd.model import load_baked_model
from vid2player3d_amd.racket import PLAYER_ARM_LIMITS, with_racket


@pytest.fixture(scope="module")
def racket_model():
    return with_racket(load_baked_model())[0]


def _fly(model, limits, tar, steps=30, dof_vel=None, solver=0):
    o = po.PhysOracle(model, po.default_params(joint_limits=int(limits), solver_type=solver))
    root = np.zeros(13)
    root[2], root[6] = 3.0, 1.0
    o.set_state(root, np.zeros(69), np.zeros(69) if dof_vel is None else dof_vel)
    for _ in range(steps):
        o.step(pd_target=tar)
    return o.get_state()


def test_player_arm_ranges_are_on_the_racket_model(racket_model):
    m = racket_model
    for name, ranges in PLAYER_ARM_LIMITS.items():
        j = 3 * (m.body_index(name) - 1)
        for i, rg in enumerate(ranges):
            if rg is not None:
                assert np.allclose(np.rad2deg([m.limit_lower[j + i], m.limit_upper[j + i]]), rg)
    base = load_baked_model()
    assert (base.limit_upper - base.limit_lower >= 2 * np.pi - 1e-6).all(), "the amass MJCF has no DOF narrower than a full turn"


@pytest.mark.parametrize("solver", [0, 1], ids=["pgs", "tgs"])
def test_drive_beyond_the_range_stops_at_the_limit(racket_model, solver):
    m = racket_model
    jw, je = 3 * (m.body_index("R_Wrist") - 1), 3 * (m.body_index("R_Elbow") - 1)
    tar = np.zeros(69)
    tar[jw:jw + 3] = [1.0, 1.2, -2.0]
    tar[je] = 2.5
    _, dp_free, _, _ = _fly(m, False, tar, solver=solver)
    _, dp_lim, _, _ = _fly(m, True, tar, solver=solver)
    assert np.rad2deg(dp_free[jw]) > 40 and np.rad2deg(dp_free[je]) > 100  # the drives do go there when nothing stops them
    lim = np.deg2rad([10.0, 45.0, -90.0])
    assert np.abs(dp_lim[jw:jw + 3] - lim).max() < 2e-3, np.rad2deg(dp_lim[jw:jw + 3])
    assert abs(dp_lim[je] - np.deg2rad(90.0)) < 2e-3


def test_rows_far_from_their_limits_change_nothing(racket_model):
    """A speculative row only acts on an approach that would cross the limit within the substep: with small targets the run with
    limits is identical to the run without."""
    m = racket_model
    rng = np.random.default_rng(3)
    tar = rng.normal(0, 0.05, size=69)
    a = _fly(m, False, tar, steps=10)
    b = _fly(m, True, tar, steps=10)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("solver", [0, 1], ids=["pgs", "tgs"])
def test_fast_approach_is_stopped_within_the_substep(racket_model, solver):
    """Wrist spun at 20 rad/s towards its 10-degree limit: the limit is reached, not crossed (speculative bias gap / h)."""
    m = racket_model
    jw = 3 * (m.body_index("R_Wrist") - 1)
    dv = np.zeros(69)
    dv[jw] = 20.0
    tar = np.zeros(69)
    tar[jw] = 1.0  # the drive keeps pushing as well
    _, dp, _, _ = _fly(m, True, tar, steps=1, dof_vel=dv, solver=solver)
    assert dp[jw] <= np.deg2rad(10.0) + 2e-3 and dp[jw] > np.deg2rad(9.0), np.rad2deg(dp[jw])
    _, dp_free, _, _ = _fly(m, False, tar, steps=1, dof_vel=dv, solver=solver)
    assert dp_free[jw] > np.deg2rad(14.0)


@pytest.mark.parametrize("solver", [0, 1], ids=["pgs", "tgs"])
def test_limits_and_contacts_share_the_sweep(racket_model, solver):
    """Lying on the ground with the wrist driven into its limit: finite, bounded, the limit holds while contacts are active."""
    m = racket_model
    o = po.PhysOracle(m, po.default_params(joint_limits=1, solver_type=solver))
    root = np.zeros(13)
    root[2], root[3:7] = 0.12, [np.sqrt(0.5), 0.0, 0.0, np.sqrt(0.5)]
    o.set_state(root, np.zeros(69), np.zeros(69))
    jw = 3 * (m.body_index("R_Wrist") - 1)
    tar = np.zeros(69)
    tar[jw] = 1.0
    touched = 0
    for _ in range(30):
        cf, _, ids = o.step(pd_target=tar)
        touched = max(touched, int((ids >= 0).any(axis=1).sum()))
    _, dp, dv, rb = o.get_state()
    assert touched >= 4 and np.isfinite(rb).all()
    assert dp[jw] < np.deg2rad(10.0) + 5e-3


def test_speculative_activation_agrees_with_rows_that_are_always_on(racket_model):
    """limit_margin: a row exists only while its DOF is within 0.05 rad of the limit or would reach it at the approach rate of v*.  Driven
    into the limits (and spun into them) the result is the one of rows that always exist (limit_margin = 1e9, the model before ABI 9):
    a row that is absent is a row that would not have acted."""
    m = racket_model
    jw, je = 3 * (m.body_index("R_Wrist") - 1), 3 * (m.body_index("R_Elbow") - 1)
    tar = np.zeros(69)
    tar[jw:jw + 3] = [1.0, 1.2, -2.0]
    tar[je] = 2.5
    dv = np.zeros(69)
    dv[jw + 1] = 15.0

    def run(margin):
        o = po.PhysOracle(m, po.default_params(joint_limits=1, limit_margin=margin))
        root = np.zeros(13)
        root[2], root[6] = 3.0, 1.0
        o.set_state(root, np.zeros(69), dv)
        out = []
        for _ in range(40):
            o.step(pd_target=tar)
            out.append(o.get_state()[1].copy())
        return np.array(out)

    a, b = run(0.05), run(1e9)
    assert np.abs(a - b).max() < 1e-6, np.abs(a - b).max()
    assert np.abs(a[-1, jw:jw + 3] - np.deg2rad([10.0, 45.0, -90.0])).max() < 2e-3
