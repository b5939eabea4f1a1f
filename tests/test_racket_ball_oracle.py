"""Racket + ball in the C oracle: physical invariants of the restatement (PhysX is closed: parity unpinned) - free
flight with drag and Magnus lift, bounce height, friction spin-up, racket hit - and the composite body model with the welded racket."""
import numpy as np
import pytest

from oracle.phys_oracle import PhysOracle, default_params
from vid2player3d_amd import racket as R
from vid2player3d_amd.model import load_baked_model

BASE = [0.5, 0.5, 0.5, 0.5]


@pytest.fixture(scope="module")
def models():
    base = load_baked_model()
    return base, R.with_racket(base)


def far_humanoid(o):
    root = np.zeros(13)
    root[0:3] = [50.0, 50.0, 5.0]
    root[3:7] = BASE
    o.set_state(root, np.zeros(69), np.zeros(69))


def test_racket_is_folded_into_the_wrist(models):
    base, (m, geom) = models
    b = base.body_index("R_Wrist")
    assert geom["racket_link"] == b == 22
    handle = 500 * np.pi * 0.016 ** 2 * 0.35
    head = 150 * np.pi * 0.15 ** 2 * np.linalg.norm([0.03, 0.03])
    assert abs(geom["racket_mass"] - (handle + head)) < 1e-12 and abs(m.mass[b] - base.mass[b] - handle - head) < 1e-12
    assert m.com[b][0] < base.com[b][0] - 0.2  # the centre of mass moves out along the racket (-x of the wrist frame)
    assert np.all(np.linalg.eigvalsh(m.inertia[b]) > 0) and np.linalg.eigvalsh(m.inertia[b]).max() > 20 * np.linalg.eigvalsh(base.inertia[b]).max()
    assert np.array_equal(np.delete(m.mass, b), np.delete(base.mass, b))
    n = np.diff(m.hull_offsets)
    assert n[b] <= 64 and np.array_equal(np.delete(n, b), np.delete(np.diff(base.hull_offsets), b))
    cyl = geom["cylinders"]
    assert np.allclose(cyl[0]["center"], [-0.175, 0, 0]) and abs(cyl[0]["half_len"] - 0.175) < 1e-12
    assert np.allclose(cyl[1]["center"], [-0.5, 0, 0]) and np.allclose(cyl[1]["axis"], np.array([0, 1, 1]) / np.sqrt(2))


def test_free_flight_drag_and_magnus(models):
    _, (m, geom) = models
    o = PhysOracle(m, default_params())
    far_humanoid(o)
    o.attach_ball(geom)
    # no spin: the force is pure drag, opposite to the velocity, kf cd |v|^2
    ball = np.zeros(13); ball[0:3] = [0, 0, 3]; ball[6] = 1; ball[7:10] = [30.0, 0, 0]
    o.set_ball(ball)
    o.step_ball(nsub=2, hold=0, sub_per_sim=2)
    v = o.get_ball()[7:10]
    kf = 1.21 * np.pi * 0.032 ** 2 / 2
    ax = -(kf * 0.55 * 30.0 ** 2) / 0.057
    assert abs((v[0] - 30.0) / (2 / 120) - ax) < 0.02 * abs(ax) and abs((v[2]) / (2 / 120) + 9.81) < 1e-4 and v[1] == 0  # (cl(v, 0) = 1 / (2 + v / 1e-6) is tiny, not zero)
    # with spin the lift points DOWN for a ball flying horizontally whatever the spin axis (the reference's formula uses the rate only)
    for spin in ([0, 200.0, 0], [0, -200.0, 0], [0, 0, 200.0]):
        ball[10:13] = spin
        o.set_ball(ball)
        o.step_ball(nsub=2, hold=0, sub_per_sim=2)
        az = o.get_ball()[9] / (2 / 120)
        cl = 1.0 / (2 + 30.0 / (200.0 / (2 * np.pi)))
        assert az < -9.81 and abs(az + 9.81 + kf * cl * 900.0 / 0.057) < 0.03 * abs(az)


@pytest.mark.parametrize("solver", [0, 1], ids=["pgs", "tgs"])
def test_bounce_height_and_friction(models, solver):
    _, (m, geom) = models
    o = PhysOracle(m, default_params(solver_type=solver))
    far_humanoid(o)
    o.attach_ball(geom)
    ball = np.zeros(13); ball[0:3] = [0, 0, 1.0]; ball[6] = 1
    o.set_ball(ball)
    zs, vzs = [], []
    for _ in range(60):
        o.step_ball(nsub=4, hold=0, sub_per_sim=2)
        b = o.get_ball()
        zs.append(b[2]); vzs.append(b[9])
    zs = np.array(zs)
    assert zs.min() > 0.032 - 0.021  # never deeper than the contact offset below touching
    vzs = np.array(vzs)
    k = int(np.argmax(vzs > 0))            # first control step after the first bounce
    k2 = k + int(np.argmax(vzs[k:] < 0))   # apex of the rebound
    rebound = zs[k:k2 + 1].max() - 0.032
    # restitution 0.5 against the ground: rebound height = e^2 x drop height (drag is ~1 % over 1 m)
    assert abs(rebound / (1.0 - 0.032) - 0.25) < 0.05, rebound  # (+- one substep of travel: the speculative contact turns the ball around up to 3.5 cm early)
    # a ball that lands with horizontal speed and no spin picks up forward spin from friction and loses horizontal speed
    ball[7:10] = [5.0, 0, 0]
    o.set_ball(ball)
    for _ in range(20):
        o.step_ball(nsub=4, hold=0, sub_per_sim=2)
    b = o.get_ball()
    assert b[7] < 5.0 - 0.3 and b[11] > 10.0  # rolling forward about +y


@pytest.mark.parametrize("solver", [0, 1], ids=["pgs", "tgs"])
def test_racket_hit_exchanges_momentum(models, solver):
    """A ball thrown at the face of the racket of a floating, limp humanoid (no gravity, no drives): it comes back (restitution 1), the
    total linear momentum of humanoid + ball is conserved, the hit is reported for that simulate() call only."""
    _, (m, geom) = models
    zeros = np.zeros(69)
    o = PhysOracle(m, default_params(gravity_z=0.0, ang_damp=0.0, solver_type=solver), kp=zeros, kd=zeros)
    root = np.zeros(13); root[2] = 3.0; root[3:7] = BASE
    o.set_state(root, zeros, zeros)
    o.attach_ball(geom, material={"ang_damp": 0.0})
    rb = o.get_state()[3]
    wrist_pos, wrist_q = rb[22, 0:3], rb[22, 3:7]
    from scipy.spatial.transform import Rotation

    Rw = Rotation.from_quat(wrist_q).as_matrix()
    centre = wrist_pos + Rw @ geom["cylinders"][1]["center"]
    normal = Rw @ geom["cylinders"][1]["axis"]
    ball = np.zeros(13); ball[6] = 1
    ball[0:3] = centre + 0.14 * normal
    ball[7:10] = -6.0 * normal  # (slow enough for the drag impulse over the test, ~5e-3 N s, not to mask the momentum balance)
    o.set_ball(ball)
    p0 = o.diagnostics()["P"] + 0.057 * ball[7:10]
    hits = []
    for _ in range(4):
        *_, per_sim, hit, bc = o.step_ball(nsub=4, hold=0, sub_per_sim=2)
        hits += hit.tolist()
    b = o.get_ball()
    vn = b[7:10] @ normal
    assert sum(hits) >= 1 and hits[-1] == 0  # polled after each simulate() on its LAST substep, like the net contact force tensor: the hit lands on one
    assert 1.5 < vn < 6.0  # bounced back; slower than it came because the 0.6 kg racket on a limp arm recoils
    p1 = o.diagnostics()["P"] + 0.057 * b[7:10]
    # of 0.34 N s: what is missing is the drag on the ball and the first-order-in-h momentum drift of the limp, now tumbling arm chain
    # (test_phys_oracle.py::test_free_flight_conserves_momentum_and_energy quantifies that drift)
    assert np.abs(p1 - p0).max() < 2e-2
    assert np.linalg.norm(o.get_state()[3][22, 7:10]) > 0.05  # the wrist was pushed


@pytest.mark.parametrize("solver", [0, 1], ids=["pgs", "tgs"])
def test_ball_bounces_off_a_link(models, solver):
    """Ball x hull contacts: a ball thrown at the chest of a floating, limp humanoid bounces off with about half its approach speed
    (restitution (1 + 0) / 2 against a link that is far heavier than the ball), total linear momentum is conserved, and the racket-hit
    flags stay clear; with body_contacts off the same ball flies through."""
    _, (m, geom) = models
    zeros = np.zeros(69)
    out = {}
    for on in (True, False):
        o = PhysOracle(m, default_params(gravity_z=0.0, ang_damp=0.0, solver_type=solver), kp=zeros, kd=zeros)
        root = np.zeros(13); root[2] = 3.0; root[3:7] = BASE
        o.set_state(root, zeros, zeros)
        o.attach_ball(geom, material={"ang_damp": 0.0}, body_contacts=on)
        rb = o.get_state()[3]
        from scipy.spatial.transform import Rotation

        b = 11  # Chest
        off = np.asarray(m.hull_offsets)
        v = np.asarray(m.hull_verts)[off[b]:off[b + 1]]
        centre = rb[b, 0:3] + Rotation.from_quat(rb[b, 3:7]).as_matrix() @ (0.5 * (v.min(0) + v.max(0)))
        d = np.array([1.0, 0.0, 0.0])  # the humanoid faces +x in this pose
        ball = np.zeros(13); ball[6] = 1
        ball[0:3] = centre + 0.5 * d
        ball[7:10] = -8.0 * d
        o.set_ball(ball)
        p0 = o.diagnostics()["P"] + 0.057 * ball[7:10]
        hits, force = [], 0.0
        for _ in range(4):
            *_, per_sim, hit, bc = o.step_ball(nsub=4, hold=0, sub_per_sim=2)
            hits += hit.tolist()
            force = max(force, np.abs(o.ball_body_force).max())
        bb = o.get_ball()
        p1 = o.diagnostics()["P"] + 0.057 * bb[7:10]
        out[on] = (bb, p0, p1, hits, o.get_state()[3][b, 7:10].copy())
    bb, p0, p1, hits, vchest = out[True]
    assert sum(hits) == 0
    assert bb[7] > 1.0 and 3.0 < np.linalg.norm(bb[7:10]) < 5.0, bb[7:10]  # came back, off an oblique surface, with ~ half of 8 m/s (friction and the limp torso take a little more)
    assert np.abs(p1 - p0).max() < 2e-2         # (drag on the ball + the first-order drift of the limp chain, see the racket test)
    assert vchest[0] < -1e-3                    # the chest was pushed back
    bb_off = out[False][0]
    assert bb_off[7] < -7.0                     # without the hull contacts the ball keeps flying (only drag)


def test_left_handed_player_is_the_mirror_image(models):
    """data/assets/smpl_mesh_humanoid_nadal.xml: the racket on L_Wrist at (+0.5, 0, 0), handle towards +x; the racket arm's ranges on the
    left elbow / wrist (Wrist_x -90 .. 10 like federer's); same racket mass."""
    base, (mr, gr) = models
    ml, gl = R.with_racket(base, player="nadal")
    b = base.body_index("L_Wrist")
    assert gl["racket_link"] == b == 17 and gl["player"] == "nadal"
    assert abs(gl["racket_mass"] - gr["racket_mass"]) < 1e-12
    assert np.allclose(gl["cylinders"][0]["center"], [0.175, 0, 0]) and np.allclose(gl["cylinders"][1]["center"], [0.5, 0, 0])
    assert np.allclose(gl["racket_offset"], [0.5, 0, 0])
    assert ml.com[b][0] > base.com[b][0] + 0.2 and np.array_equal(ml.mass[22], base.mass[22])
    j = 3 * (b - 1)
    assert np.allclose(np.rad2deg(ml.limit_lower[j:j + 3]), [-90, -45, -90]) and np.allclose(np.rad2deg(ml.limit_upper[j:j + 3]), [10, 45, 90])
    je = 3 * (base.body_index("L_Elbow") - 1)
    assert np.allclose(np.rad2deg([ml.limit_lower[je], ml.limit_upper[je]]), [-180, 90])
    jr = 3 * (22 - 1)
    assert np.all(ml.limit_upper[jr:jr + 3] - ml.limit_lower[jr:jr + 3] >= 6.28)   # the right arm is free
    mf, _ = R.with_racket(base, player="federer")
    assert np.allclose(np.rad2deg([mf.limit_lower[jr], mf.limit_upper[jr]]), [-90, 10])


def test_racket_model_inherits_gain_scales_and_refuses_a_second_racket():
    """cfg env kp_scale / kd_scale / default_humanoid_mass reach the racket model (the task takes its PD gains from it), and a model that
    already carries a racket is not folded again (mass, inertia and rim vertices would double)."""
    from vid2player3d_amd.model import load_baked_model

    plain = load_baked_model()
    scaled = load_baked_model(default_humanoid_mass=75.0, kp_scale=1.5, kd_scale=0.5)
    m0, _ = R.with_racket(plain)
    m1, _ = R.with_racket(scaled)
    r = m1.total_mass / m0.total_mass  # (same bodies: 1)
    assert abs(r - 1.0) < 1e-12
    assert np.allclose(m1.kp, m0.kp * 1.5 * 90.0 / 75.0) and np.allclose(m1.kd, m0.kd * 0.5 * 90.0 / 75.0)
    with pytest.raises(ValueError):
        R.with_racket(m1)
    assert np.allclose(scaled.scaled(1.1).kp / plain.scaled(1.1).kp, 1.5 * 90.0 / 75.0)


def test_ball_sensitivity_leaves_the_states_alone(models):
    """PhysOracle.ball_sensitivity (the conditioning term of the GPU parity bounds with a ball): humanoid and ball states are restored
    bit for bit, the step taken afterwards is the unperturbed one, and a standing humanoid with a ball in flight is well conditioned
    in the ball (free flight) and has a finite, positive figure on the links in contact."""
    _, (m, geom) = models
    o = PhysOracle(m, default_params(), kp=m.kp.astype(np.float32), kd=m.kd.astype(np.float32))
    root = np.zeros(13); root[2] = 0.93; root[3:7] = BASE
    o.set_state(root, np.zeros(69), np.zeros(69))
    o.attach_ball(geom)
    ball = np.zeros(13); ball[0:3] = [2.0, 0, 1.5]; ball[6] = 1; ball[7:10] = [-10.0, 0, 1.0]
    o.set_ball(ball)
    for _ in range(6):  # settle onto the feet
        o.step_ball(pd_target=np.zeros(69), nsub=4, hold=0, sub_per_sim=2)
    s0, b0 = o.get_state(), o.get_ball()
    sens = o.ball_sensitivity(pd_target=np.zeros(69), nsub=4, hold=0, sub_per_sim=2)
    for x, y in zip(s0, o.get_state()):
        assert np.array_equal(x, y)
    assert np.array_equal(b0, o.get_ball())
    twin = PhysOracle(m, default_params(), kp=m.kp.astype(np.float32), kd=m.kd.astype(np.float32))
    twin.set_state(s0[0], s0[1], s0[2])
    twin.attach_ball(geom)
    twin.set_ball(b0)
    a = o.step_ball(pd_target=np.zeros(69), nsub=4, hold=0, sub_per_sim=2)
    b = twin.step_ball(pd_target=np.zeros(69), nsub=4, hold=0, sub_per_sim=2)
    assert np.allclose(a[0], b[0], atol=1e-9) and np.allclose(a[3], b[3], atol=1e-12)
    assert sens["rb"].shape == (24, 13) and sens["ball"].shape == (2, 13) and sens["bc"].shape == (3, 3)
    assert 0 < sens["ball"][:, 7:10].max() < 1e-4 and 0 < sens["rb"][:, 7:13].max() < 1.0 and np.isfinite(sens["cf"]).all()
