"""Physical-invariant checks of the C physics oracle (oracle/phys).  Parity with PhysX is unpinned
(closed binary); these tests establish that the restatement is a self-consistent rigid-body model."""
import numpy as np
import pytest

from oracle.phys_oracle import PhysOracle, default_params
from vid2player3d_amd.model import load_baked_model

BASE = np.array([0.5, 0.5, 0.5, 0.5])


@pytest.fixture(scope="module")
def model():
    return load_baked_model()


def standing_state(rng, height=0.95, pose_sigma=0.3, vel_sigma=1.0):
    root = np.zeros(13)
    root[2] = height
    root[3:7] = BASE
    root[7:13] = rng.normal(0, vel_sigma, 6)
    return root, rng.normal(0, pose_sigma, 69), rng.normal(0, vel_sigma, 69)


def _free_flight_drift(model, h, t_end=0.2):
    rng = np.random.default_rng(0)
    zeros = np.zeros(69)
    o = PhysOracle(model, default_params(h=h, enable_contact=False, gravity_z=0.0, ang_damp=0.0), kp=zeros, kd=zeros, armature=zeros)
    root, dp, dv = standing_state(rng, height=3.0)
    o.set_state(root, dp, dv)
    d0 = o.diagnostics()
    for _ in range(int(round(t_end / h))):
        o.step(nsub=1, hold=0)
    d1 = o.diagnostics()
    return (np.abs(d1["P"] - d0["P"]).max() / np.abs(d0["P"]).max(), np.abs(d1["L"] - d0["L"]).max() / np.abs(d0["L"]).max(),
            abs(d1["ke"] - d0["ke"]) / d0["ke"])


def test_free_flight_conserves_momentum_and_energy(model):
    """Torque-free, gravity-free flight: linear/angular momentum and kinetic energy are conserved by the
    continuous model, so the discrete drift must be small AND first order in h (a wrong Coriolis/bias
    term would leave an h-independent residual)."""
    coarse = _free_flight_drift(model, 1e-3)
    fine = _free_flight_drift(model, 2.5e-4)
    for c, f in zip(coarse, fine):
        assert c < 2e-2
        assert f < 0.4 * c, (coarse, fine)


def test_free_fall_momentum_rate_is_weight(model):
    rng = np.random.default_rng(1)
    zeros = np.zeros(69)
    o = PhysOracle(model, default_params(h=1.0 / 120, enable_contact=False, ang_damp=0.0), kp=zeros, kd=zeros, armature=zeros)
    root, dp, dv = standing_state(rng, height=5.0)
    root[10:13] = 0.0
    o.set_state(root, dp, 0.0 * dv)   # no internal motion: the discrete update is exact
    d0 = o.diagnostics()
    n = 30
    for _ in range(n):
        o.step(nsub=1, hold=0)
    d1 = o.diagnostics()
    expect = d0["P"] + np.array([0, 0, -9.81 * model.total_mass * n / 120.0])
    assert np.allclose(d1["P"], expect, atol=1e-6 * model.total_mass)


def test_energy_is_dissipated_by_drives(model):
    """With PD drives holding the current pose (target = q) and no gravity, kinetic energy can only fall."""
    rng = np.random.default_rng(2)
    o = PhysOracle(model, default_params(enable_contact=False, gravity_z=0.0))
    root, dp, dv = standing_state(rng, height=3.0, vel_sigma=2.0)
    o.set_state(root, dp, dv)
    ke = [o.diagnostics()["ke"]]
    for _ in range(20):
        o.step(pd_target=o.get_state()[1], nsub=1, hold=0)
        ke.append(o.diagnostics()["ke"])
    assert all(b <= a * (1 + 1e-9) for a, b in zip(ke, ke[1:]))
    assert ke[-1] < ke[0]
    assert np.abs(o.get_state()[2]).max() < 0.25 * np.abs(dv).max()   # joint rates are damped out; free-root motion remains


def test_external_wrench_changes_momentum(model):
    zeros = np.zeros(69)
    o = PhysOracle(model, default_params(enable_contact=False, gravity_z=0.0, ang_damp=0.0), kp=zeros, kd=zeros, armature=zeros)
    root = np.zeros(13)
    root[2] = 2.0
    root[3:7] = BASE
    o.set_state(root, np.zeros(69), np.zeros(69))
    f = np.array([10.0, -20.0, 30.0])
    o.step(ext_force=f, ext_torque=np.zeros(3), nsub=4, hold=2)
    assert np.allclose(o.diagnostics()["P"], f * 2 / 120.0, rtol=1e-3)


def test_rest_contact_supports_weight(model):
    """Standing on the plane with the drives holding the rest pose: the feet carry the body weight while
    it is upright; it eventually tips over (no balance controller) and then rests on the ground, again
    carried by the contacts, without tunnelling."""
    o = PhysOracle(model)
    root = np.zeros(13)
    root[2] = 0.965
    root[3:7] = BASE
    pose = np.zeros(69)
    o.set_state(root, pose, np.zeros(69))
    weight = 9.81 * model.total_mass
    fz = []
    for _ in range(30):
        cf, _, ids = o.step(pd_target=pose, nsub=4, hold=0)
        fz.append(cf[:, 2].sum())
    rb = o.get_state()[3]
    assert rb[13, 2] > 1.5                                   # still upright after 1 s
    assert abs(np.mean(fz[-15:]) - weight) < 0.1 * weight
    assert set(np.nonzero((ids >= 0).any(axis=1))[0].tolist()) <= {3, 4, 7, 8}   # only feet touch
    for _ in range(100):
        cf, _, ids = o.step(pd_target=pose, nsub=4, hold=0)
        fz.append(cf[:, 2].sum())
    rb = o.get_state()[3]
    assert rb[13, 2] < 0.4                                   # fell over
    assert rb[:, 2].min() > -0.02                            # nothing tunnels the plane
    assert abs(np.mean(fz[-20:]) - weight) < 0.1 * weight


def test_contact_manifold_is_bounded_and_deterministic(model):
    o = PhysOracle(model)
    root = np.zeros(13)
    root[2] = 0.12                                     # lying on its back/side: many bodies touch
    root[3:7] = [0.0, 0.0, 0.0, 1.0]
    o.set_state(root, np.zeros(69), np.zeros(69))
    _, _, ids = o.step(pd_target=np.zeros(69), nsub=1, hold=0)
    o.set_state(root, np.zeros(69), np.zeros(69))
    _, _, ids2 = o.step(pd_target=np.zeros(69), nsub=1, hold=0)
    assert np.array_equal(ids, ids2)
    assert (ids >= 0).sum() > 12
    for b in range(24):
        sel = ids[b][ids[b] >= 0]
        assert len(set(sel.tolist())) == len(sel)
        assert all(v // 64 == b for v in sel)


def test_sensitivity_leaves_the_states_alone_and_finds_the_ill_conditioned_envs():
    """BatchOracle.sensitivity (the conditioning term of the GPU parity bounds): the batch's states are untouched (the step that follows
    equals the step without it, bit for bit), the result is reproducible, and on contact-rich states it spreads over orders of magnitude -
    the step map of the model amplifies float32-rounding perturbations 10 x in the median env and 1000 x in a few (tools/gain_probe.py)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle.phys_oracle import BatchOracle, default_params
    from tools.gain_probe import fixture

    n = 96
    bm, root, dpos, dvel, pd, force, torque = fixture(n, seed=3, lift=-0.75, vel_sigma=0.2)   # fallen: ~8 touched links per env
    a = BatchOracle(bm, n, default_params())
    a.set_state(root, dpos, dvel)
    s1 = a.sensitivity(pd, force, torque, seed=5)
    ra = a.step(pd, force, torque)
    b = BatchOracle(bm, n, default_params())
    b.set_state(root, dpos, dvel)
    rb = b.step(pd, force, torque)
    for k in ("dvel", "rb", "cf", "root", "dpos"):
        assert np.array_equal(ra[k], rb[k]), k
    c = BatchOracle(bm, n, default_params())
    c.set_state(root, dpos, dvel)
    s2 = c.sensitivity(pd, force, torque, seed=5)
    assert np.array_equal(s1["dvel"], s2["dvel"]) and (s1["dvel"] >= 0).all() and np.isfinite(s1["cf"]).all()
    gain = s1["dvel"].max(axis=1) / 1e-6
    assert np.median(gain) > 3.0 and gain.max() > 10.0 * np.median(gain), (np.median(gain), gain.max())
    # without contacts the step is well conditioned everywhere
    d = BatchOracle(bm, n, default_params(enable_contact=0))
    d.set_state(root + np.array([0, 0, 2.0] + [0] * 10), dpos, dvel)
    g0 = d.sensitivity(pd, force, torque, seed=5)["dvel"].max(axis=1) / 1e-6
    assert g0.max() < 40.0, g0.max()


def test_alternating_sweeps_would_cost_accuracy():
    """Why the PGS sweeps all ascend (round 4): sweeps in alternating direction (v2p_oracle_experiment(4)) would let the
    engine's tree walk skip the trip from the last touched link back to the first - 8 % fewer instructions - but the link a sweep ends on
    is then solved twice in a row, and the model's 4 sweeps end farther from the converged solution: in the standing states where the two
    orders differ at all (about half of them) by a factor of 1.4 (geometric mean), in three of four of those states, +18 % in the mean over
    all (measured on the envs in which both orders converge to the same solution)."""
    import ctypes as C
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle.phys_oracle import BatchOracle, default_params, lib
    from tools.gain_probe import fixture

    n = 512
    bm, root, dpos, dvel, pd, force, torque = fixture(n, seed=11, lift=0.0, vel_sigma=0.5)

    def run(n_iter, alternate):
        lib().v2p_oracle_experiment(C.c_int(4 if alternate else 0))
        try:
            p = default_params()
            p.n_iter = n_iter
            o = BatchOracle(bm, n, p)
            o.set_state(root, dpos, dvel)
            return o.step(pd, force, torque, nsub=1, hold=1)["dvel"]
        finally:
            lib().v2p_oracle_experiment(C.c_int(0))

    conv_f, conv_a = run(600, False), run(600, True)
    same = np.abs(conv_f - conv_a).max(axis=1) < 1e-6   # box friction bounded by the current normal impulse: not every env converges
    assert same.mean() > 0.6
    res_f = np.abs(run(4, False) - conv_f).max(axis=1)[same]
    res_a = np.abs(run(4, True) - conv_a).max(axis=1)[same]
    differ = (res_a != res_f) & (res_a > 0) & (res_f > 0)
    ratio = float(np.exp(np.mean(np.log(res_a[differ] / res_f[differ]))))
    worse = float((res_a[differ] > res_f[differ]).mean())
    print("[sweeps] distance to the converged solution after 4 sweeps over %d envs: mean %.3f (ascending) vs %.3f rad/s (alternating); the orders differ in "
          "%d envs, there alternating is the farther one in %.0f %%, geometric mean of the ratio %.2f" % (same.sum(), res_f.mean(), res_a.mean(), differ.sum(), 100 * worse, ratio))
    assert differ.mean() > 0.3 and worse > 0.6 and ratio > 1.2 and res_a.mean() > 1.1 * res_f.mean()


# ------------------------------------------------------------------------------------------------------------------------------------
# Closed-form facts of the contact / drive model (round 5): what a PhysX trace would be compared on first.
def box_model(model, half=(0.25, 0.25, 0.1), mass=10.0):
    """A rigid box on the plane inside the oracle's 24-link format: the root link IS the box (8 hull vertices, uniform density), the
    other 23 links are 0.1 g points stacked above it on stiff drives, far from the ground."""
    from vid2player3d_amd.model import BodyModel

    hx, hy, hz = half
    blob = dict(model.blob)
    nb = 24
    corners = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    blob["hull_verts"] = np.concatenate([corners, np.zeros((nb - 1, 3))])
    blob["hull_offsets"] = np.concatenate([[0, 8], 8 + np.arange(1, nb)]).astype(np.int32)
    m = np.full(nb, 1e-4)
    m[0] = mass
    inertia = np.tile(np.eye(3) * 1e-8, (nb, 1, 1))
    inertia[0] = np.diag([mass / 3.0 * (hy * hy + hz * hz), mass / 3.0 * (hx * hx + hz * hz), mass / 3.0 * (hx * hx + hy * hy)])
    lp = np.tile(np.array([0.0, 0.0, 0.05]), (nb, 1))
    lp[0] = 0.0
    blob.update(mass=m, com=np.zeros((nb, 3)), inertia=inertia, local_pos=lp, kp=np.full(69, 50.0), kd=np.full(69, 5.0), armature=np.full(69, 1e-3))
    return BodyModel(blob, default_humanoid_mass=float(m.sum()))  # (gain scale 1)


def _push_box(model, alpha, direction, steps=30, mu=1.0, friction_frame=0):
    """The box at rest on the plane, pushed horizontally at its centre of mass with alpha x its weight along `direction` for `steps`
    control steps: a slope of tan(theta) = alpha in the frame of the plane.  Returns the box's horizontal velocity after every step."""
    bm = box_model(model)
    o = PhysOracle(bm, default_params(mu=mu, ang_damp=0.0, friction_frame=friction_frame))
    root = np.zeros(13)
    root[2] = 0.1
    root[6] = 1.0  # identity: the box's z is the world's
    o.set_state(root, np.zeros(69), np.zeros(69))
    for _ in range(10):
        o.step(pd_target=np.zeros(69), nsub=4, hold=0)  # settle
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    f = alpha * bm.total_mass * 9.81 * d
    v = []
    for _ in range(steps):
        cf, _, ids = o.step(pd_target=np.zeros(69), ext_force=f, ext_torque=np.zeros(3), nsub=4, hold=4)
        v.append(o.get_state()[0][7:10].copy())
    assert (ids[0] >= 0).sum() == 4 and (ids[1:] >= 0).sum() == 0  # the box rests on its four bottom corners, nothing else touches
    assert abs(cf[:, 2].sum() - bm.total_mass * 9.81) < 0.02 * bm.total_mass * 9.81
    return np.array(v), d


def test_box_on_a_slope_sticks_below_the_friction_angle_and_slides_above(model):
    """mu = 1: a box on a slope of angle theta (here: a horizontal push of tan(theta) x its weight on a level plane) stays put below 45
    degrees and accelerates with (tan(theta) - mu) g cos-free above.  Along a tangent axis of the contact frame (world x) the threshold is
    mu exactly."""
    v, d = _push_box(model, 0.9, [1, 0, 0])
    assert np.abs(v[-10:]).max() < 2e-3, np.abs(v[-10:]).max()          # 42 degrees: sticks
    v, d = _push_box(model, 1.1, [1, 0, 0])
    t = len(v) / 30.0
    assert abs(v[-1] @ d - 0.1 * 9.81 * t) < 0.05 * 0.1 * 9.81 * t, (v[-1], 0.1 * 9.81 * t)  # 47.7 degrees: slides with (alpha - mu) g
    assert np.all(np.diff(v @ d) > 0)
    v, _ = _push_box(model, 0.45, [1, 0, 0], mu=0.5)                      # the threshold follows mu
    assert np.abs(v[-10:]).max() < 2e-3
    v, d = _push_box(model, 0.55, [1, 0, 0], mu=0.5)
    assert abs(v[-1] @ d - 0.05 * 9.81 * len(v) / 30.0) < 0.05 * 0.05 * 9.81 * len(v) / 30.0


def test_friction_limit_is_a_pyramid_aligned_with_the_world_axes(model):
    """The rows t1, t2 of a ground contact are world x and y, each clamped to mu x the normal impulse on its own (box friction): along the
    DIAGONAL the box holds up to sqrt(2) mu - it still sticks at alpha = 1.3 (where a friction cone would let it go: 1.3 > mu) and slides
    at 1.5 with (alpha - sqrt(2) mu) g.  This is the model's friction law, stated; PhysX's patch friction is closer to a cone - one of the
    first things an Isaac Gym trace would show."""
    v, d = _push_box(model, 1.3, [1, 1, 0])
    # (sticks; what is left is the creep of four unconverged sweeps, 3 mm/s and 0.0005 g - a cone would let the box go with 0.3 g)
    assert np.abs(v[-10:]).max() < 1e-2 and abs((v[-1] - v[-11]) @ d) / (10 / 30.0) < 0.005 * 9.81, (np.abs(v[-10:]).max(), (v[-1] - v[-11]) @ d)
    v, d = _push_box(model, 1.5, [1, 1, 0])
    t = len(v) / 30.0
    want = (1.5 - np.sqrt(2.0)) * 9.81 * t
    assert abs(v[-1] @ d - want) < 0.06 * want, (v[-1] @ d, want)
    assert abs(v[-1][0] - v[-1][1]) < 1e-3 * v[-1][0]  # symmetric in x and y (Gauss-Seidel visits t1 before t2: 1e-4 relative)


def test_velocity_aligned_friction_frame_is_isotropic(model):
    """v2p_oparams.friction_frame = 1 (v2p_sim_cfg.friction_frame "velocity"): t1 of a hull x ground point lies along the tangential velocity
    the point has under v*, so the friction limit along the direction of sliding is mu whatever that direction is: pushed along the DIAGONAL
    the box now lets go above mu - it slides at alpha = 1.3 with (alpha - mu) g, where the world-aligned box (friction_frame 0) still holds -
    and sticks at 0.9; along x nothing changes.  (The switch exists so that the first Isaac Gym trace can choose, tools/replay_trace.py.)"""
    g = 9.81
    for direction in ([1, 1, 0], [1, 0, 0], [0.3, -1.0, 0]):
        v, d = _push_box(model, 0.9, direction, friction_frame=1)
        assert np.abs(v[-10:]).max() < 1e-2, (direction, np.abs(v[-10:]).max())              # below mu: sticks, in every direction
        v, d = _push_box(model, 1.3, direction, friction_frame=1)
        t = len(v) / 30.0
        want = 0.3 * g * t
        assert abs(v[-1] @ d - want) < 0.06 * want, (direction, v[-1] @ d, want)              # above mu: (alpha - mu) g along the push
        assert np.linalg.norm(v[-1][:2] - (v[-1] @ d) * d[:2]) < 0.02 * want                  # ... and only along it
    v0, d = _push_box(model, 1.3, [1, 1, 0], friction_frame=0)
    assert np.abs(v0[-10:]).max() < 1e-2                                                      # the world-aligned box holds the same push


def test_ball_bounces_above_the_threshold_velocity_only():
    """sim.physx.bounce_threshold_velocity = 0.2 m/s: a ball that reaches the ground faster than that leaves it with restitution x its
    approach speed (Newton), a slower one does not bounce at all (the row only stops it)."""
    from vid2player3d_amd.racket import BALL, BALL_MATERIAL, with_racket

    m, geom = with_racket(load_baked_model())
    R, e = BALL["radius"], BALL_MATERIAL["rest_ground"]
    assert BALL_MATERIAL["bounce_threshold"] == 0.2 and e == 0.5

    def drop(height_above, v0):
        o = PhysOracle(m, default_params())
        root = np.zeros(13); root[0] = 50.0; root[2] = 5.0; root[3:7] = BASE   # the humanoid: far away, in the air
        o.set_state(root, np.zeros(69), np.zeros(69))
        o.attach_ball(geom)
        ball = np.zeros(13); ball[2] = R + height_above; ball[6] = 1.0; ball[9] = v0
        o.set_ball(ball)
        vz = [v0]
        for _ in range(40):
            o.step_ball(nsub=1, hold=0, sub_per_sim=1)
            vz.append(o.get_ball()[9])
        return np.array(vz)

    h = 1.0 / 120.0
    vz = drop(0.05, -1.0)                     # arrives at ~1.4 m/s
    k = int(np.argmax(vz > 0))                # the substep that turned it around
    approach = vz[k - 1] - 9.81 * h           # the unconstrained velocity of that substep (drag: 1e-4 m/s)
    assert approach < -0.2 and abs(vz[k] - e * (-approach)) < 0.01 * abs(approach), (vz[k], approach)
    vz = drop(0.0005, -0.05)                  # arrives at ~0.13 m/s: below the threshold
    assert vz.max() < 1e-9 and abs(vz[-1]) < 1e-6, (vz.max(), vz[-1])   # stopped, never moves up


def test_single_pd_joint_follows_the_implicit_euler_closed_form(model):
    """One drive in isolation (L_Toe about its x axis; the root weighs 10^5 kg, every other drive - the toe's own y and z included - is 10^5 .. 10^7 times stiffer, no gravity): the
    joint follows  v+ = (I' v + h kp (target - q)) / (I' + h kd + h^2 kp),  q+ = q + h v+,  I' = the link's inertia about the joint axis +
    armature - backward Euler on  I' q'' = kp (target - q) - kd q'  - substep after substep, over- and under-damped."""
    from vid2player3d_amd.model import BodyModel

    b, jx = 4, 3 * (4 - 1)   # L_Toe (a leaf), its x DOF
    for kp, kd in ((200.0, 20.0), (800.0, 0.5), (5.0, 0.0)):
        blob = dict(model.blob)
        mass = model.mass.copy(); mass[0] = 1e5
        inertia = model.inertia.copy(); inertia[0] = np.eye(3) * 1e5
        kps, kds = np.full(69, 1e8), np.full(69, 1e6)
        kps[jx], kds[jx] = kp, kd   # (the oracle takes per-axis gains: the toe's y and z stay stiff, its products of inertia cannot tilt the axis)
        blob.update(mass=mass, inertia=inertia, kp=kps, kd=kds, armature=np.full(69, 0.02))
        bm = BodyModel(blob, default_humanoid_mass=float(mass.sum()))
        c = bm.com[b]
        I_axis = bm.inertia[b][0, 0] + bm.mass[b] * (c[1] ** 2 + c[2] ** 2) + 0.02
        o = PhysOracle(bm, default_params(enable_contact=False, gravity_z=0.0, ang_damp=0.0))
        root = np.zeros(13); root[2] = 2.0; root[3:7] = BASE
        q0, w0, tar = 0.3, -1.0, np.zeros(69)
        tar[jx] = -0.2
        dp, dv = np.zeros(69), np.zeros(69)
        dp[jx], dv[jx] = q0, w0
        o.set_state(root, dp, dv)
        h, q, w = 1.0 / 120.0, q0, w0
        for k in range(40):
            o.step(pd_target=tar, nsub=1, hold=0)
            w = (I_axis * w + h * kp * (tar[jx] - q)) / (I_axis + h * kd + h * h * kp)
            q = q + h * w
            _, dpo, dvo, _ = o.get_state()
            assert abs(dvo[jx] - w) < 2e-5 * max(1.0, abs(w)) and abs(dpo[jx] - q) < 2e-6, (kp, kd, k, dvo[jx], w, dpo[jx], q)
            assert np.abs(np.delete(dpo, jx)).max() < 1e-6  # nothing else moves
