"""Racket + ball on the GPU: the HIP kernel against the C oracle on every env - ball in free flight with drag and
Magnus lift, bouncing on the ground, hit by the racket of a moving humanoid - plus the flags the reference derives per simulate() call."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from oracle import task_oracle as O
from oracle.phys_oracle import PhysOracle, default_params
from tests.gpu_util import DEV, N, T, close, synth_tables

pytestmark = pytest.mark.gpu


TIE_TOL = 5e-5  # metres: a selection of the kernel that differs from the oracle's own must be a tie of the rule (test_gpu_physics.py)


def make_rb_task(n, lib, sim_overrides=None, **env):
    from vid2player3d_amd.tasks import HumanoidSMPLIMRacketBall, default_cfg

    env.setdefault("debug_contacts", 2)  # the contact vertices of every substep: the oracle is teacher-forced with them
    env.setdefault("body_shape_mismatch", "ignore")
    env.setdefault("contact_forces_sum", True)
    cfg = default_cfg(n, motion_lib=lib, sample_first_motions=True, **env)
    cfg["sim"].update(sim_overrides or {})
    return HumanoidSMPLIMRacketBall(cfg, device_type="cuda", device_id=0)


@pytest.fixture(scope="module")
def mlib():
    from vid2player3d_amd.motion_lib import MotionLib

    return MotionLib(synth_tables(seed=5, num_clips=8, min_frames=60, max_frames=120), DEV)


def _launch(task, rng, mode):
    """Ball states [n,13] for the scenario `mode`; 'hit': aimed at the face of each env's racket head."""
    n = task.num_envs
    ball = np.zeros((n, 13), np.float32)
    ball[:, 6] = 1
    rb = N(task._rigid_body_state).reshape(n, 24, 13)
    if mode == "flight":
        ball[:, 0:3] = rb[:, 0, 0:3] + rng.uniform(-3, 3, (n, 3)) + [0, 0, 4]
        ball[:, 7:10] = rng.normal(0, 15, (n, 3))
        ball[:, 10:13] = rng.normal(0, 150, (n, 3))
    elif mode == "ground":
        ball[:, 0:2] = rb[:, 0, 0:2] + rng.uniform(3, 5, (n, 2))
        ball[:, 2] = rng.uniform(0.03, 0.25, n)
        ball[:, 7:10] = np.stack([rng.normal(0, 6, n), rng.normal(0, 6, n), rng.uniform(-12, 1, n)], 1)
        ball[:, 10:13] = rng.normal(0, 60, (n, 3))
    elif mode == "serve":
        # bench.py's serve (a ball flying at the player at ~22 m/s with top spin), started 0.6 .. 2 m in front of the root so that it
        # arrives within the steps of the test: racket, link hulls, ground - whatever is in the way
        dist = rng.uniform(0.6, 2.0, n)
        ball[:, 0:3] = rb[:, 0, 0:3] + np.stack([dist, rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.5, n)], 1)
        ball[:, 7:10] = np.array([-22.0, 0.0, 4.0]) + (rng.uniform(size=(n, 3)) - 0.5) * np.array([6.0, 3.0, 3.0])
        ball[:, 10:13] = np.array([0.0, -150.0, 0.0])
    elif mode == "joint":
        # aimed at a JOINT of each env (knee, elbow, neck ...): the hulls of the two links that meet there are both within reach of a
        # fast ball, so two hull points are active at once (one per overlapping link)
        for e in range(n):
            b = int(rng.choice([2, 6, 10, 12, 16, 3, 7]))
            target = rb[e, b, 0:3]
            d = rng.normal(size=3)
            d[2] = abs(d[2])
            d /= np.linalg.norm(d)
            speed = rng.uniform(15, 35)
            ball[e, 0:3] = target + rng.uniform(0.15, 0.3) * d
            ball[e, 7:10] = -speed * d + rb[e, b, 7:10]
            ball[e, 10:13] = rng.normal(0, 80, 3)
    elif mode == "body":
        # aimed at the hull of a link of each env (not the racket's), from a random direction
        bm = task.body_model
        off = np.asarray(bm.hull_offsets)
        hv = np.asarray(bm.hull_verts, dtype=np.float64)
        for e in range(n):
            b = int(rng.choice([0, 1, 2, 5, 6, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 20, 21, 23]))
            Rw = Rotation.from_quat(rb[e, b, 3:7]).as_matrix()
            v = hv[off[b]:off[b + 1]]
            target = rb[e, b, 0:3] + Rw @ (0.5 * (v.min(0) + v.max(0)))
            d = rng.normal(size=3)
            d[2] = abs(d[2])
            d /= np.linalg.norm(d)
            speed = rng.uniform(4, 30)
            ball[e, 0:3] = target + rng.uniform(0.2, 0.45) * d
            ball[e, 7:10] = -speed * d + rng.normal(0, 1, 3) + rb[e, b, 7:10]
            ball[e, 10:13] = rng.normal(0, 80, 3)
    else:
        geom = task.racket_geometry
        rl = geom["racket_link"]
        for e in range(n):
            Rw = Rotation.from_quat(rb[e, rl, 3:7]).as_matrix()
            centre = rb[e, rl, 0:3] + Rw @ geom["cylinders"][1]["center"]
            normal = Rw @ geom["cylinders"][1]["axis"] * (1 if e % 2 else -1)
            side = np.cross(normal, [0.3, 0.5, 0.8])
            side /= np.linalg.norm(side)
            speed = rng.uniform(4, 30)
            ball[e, 0:3] = centre + rng.uniform(0.06, 0.25) * normal + rng.uniform(0, 0.11) * side
            ball[e, 7:10] = -speed * normal + rng.normal(0, 2, 3)
            ball[e, 10:13] = rng.normal(0, 80, 3)
    return ball


@pytest.mark.parametrize("solver", ["pgs", "tgs"])
@pytest.mark.parametrize("mode,lift,limits,player", [("flight", 0.0, False, "djokovic"), ("ground", 0.0, False, "djokovic"), ("hit", 0.4, False, "djokovic"),
                                                     ("hit", 0.0, False, "djokovic"), ("hit", 0.0, True, "djokovic"), ("body", 0.4, False, "djokovic"),
                                                     ("body", 0.0, True, "federer"), ("hit", 0.0, True, "nadal"), ("joint", 0.4, False, "djokovic")])
def test_ball_step_matches_oracle(mlib, mode, lift, limits, player, solver):
    """limits: with the joint ranges of the player MJCF's racket arm enforced (v2p_sim_cfg.joint_limits) - the wrist's limit rows, its
    hull points and the ball x racket rows then all belong to the same link.  player: the asset (nadal = left-handed: racket on L_Wrist).
    solver: tgs = the solver the reference's tennis yamls name (vid2player/cfg/im/tennis_im.yaml:39, embodied_pose/cfg/djokovic_im.yaml:41)."""
    _ball_step_vs_oracle(mlib, mode, lift, limits, player, solver=solver)


@pytest.mark.parametrize("solver", ["pgs", "tgs"])
@pytest.mark.parametrize("mode,limits", [("hit", True), ("body", False), ("ground", False)])
def test_ball_step_with_the_lds_parked_build(mlib, mode, limits, solver):
    """kernel_build=1: the build that full-size batches run (the engine gives 32-env fixtures the register build, which is what every
    other small test of this file sees)."""
    _ball_step_vs_oracle(mlib, mode, 0.0, limits, "djokovic", kernel_build=1, solver=solver)


@pytest.mark.parametrize("solver", ["pgs", "tgs"])
@pytest.mark.parametrize("mode", ["hit", "body"])
def test_ball_step_with_one_body_shape_per_clip(mlib, mode, solver):
    """racket + ball on per-clip body shapes (three differently scaled bodies, each with the racket folded into its wrist): every env
    against the oracle of ITS shape"""
    from vid2player3d_amd.model import load_baked_model

    base = load_baked_model()
    _ball_step_vs_oracle(mlib, mode, 0.0, True, "djokovic", shapes=[base.scaled(0.9), base, base.scaled(1.12)], solver=solver)


@pytest.mark.parametrize("solver", ["pgs", "tgs"])
@pytest.mark.parametrize("mode", ["hit", "ground"])
def test_ball_step_with_six_substeps_per_simulate_call(mlib, mode, solver):
    """sim.substeps 6 (vid2player/cfg/controller/*.yaml: 12 substeps of 1/360 s per control step): the racket-hit poll is off then
    (humanoid_smpl_im_mvae.py:769), the bounce test uses 6 ball radii (:733); kernel vs oracle on every env, through substep jobs."""
    _ball_step_vs_oracle(mlib, mode, 0.0, True, "djokovic", substeps=6, solver=solver)


@pytest.mark.parametrize("solver", ["pgs", "tgs"])
@pytest.mark.parametrize("mode", ["hit", "body"])
def test_ball_step_with_the_velocity_aligned_friction_frame(mlib, mode, solver):
    """v2p_sim_cfg.friction_frame = velocity in the racket + ball instantiations (limit rows on): the feet's friction rows turn with their
    sliding direction, the ball's rows keep the basis of their normal; kernel vs oracle on every env."""
    _ball_step_vs_oracle(mlib, mode, 0.0, True, "djokovic", solver=solver, friction_frame="velocity")


def _ball_step_vs_oracle(mlib, mode, lift, limits, player, shapes=None, substeps=2, n=32, subset=None, steps=2, solver="pgs", **env):
    """subset: the envs that get an oracle (all by default); every comparison is restricted to them."""
    env = dict(env, contact_solver=solver)
    sub = np.arange(n) if subset is None else np.asarray(sorted(int(i) for i in subset))
    rng = np.random.default_rng({"flight": 1, "ground": 2, "hit": 3, "body": 4, "joint": 5, "serve": 6}[mode] + int(10 * lift))
    extra = dict(env, **({} if shapes is None else {"body_model": shapes, "motion_shape_ids": np.arange(8) % len(shapes)}))
    task = make_rb_task(n, mlib, joint_limits=limits, player=player, sim_overrides={"substeps": substeps}, **extra)
    rl = task.racket_geometry["racket_link"]
    assert rl == (17 if player == "nadal" else 22) and task.contact_solver == solver
    task.reset_with_times(None, T(rng.uniform(0.1, 1.0, size=n)))
    root = N(task._humanoid_root_states).copy()
    root[:, 2] += lift
    root[:, 7:13] += rng.normal(0, 0.5, (n, 6)).astype(np.float32)
    dpos = N(task._dof_pos).copy() + rng.normal(0, 0.05, (n, 69)).astype(np.float32)
    dvel = N(task._dof_vel).copy() + rng.normal(0, 1.0, (n, 69)).astype(np.float32)
    task._humanoid_root_states[:] = T(root)
    task._dof_pos[:] = T(dpos)
    task._dof_vel[:] = T(dvel)
    task._reset_env_tensors(None)
    # one physics-free refresh of the rigid-body state for the launch geometry: FK of the pushed state through the oracle
    bm = task.body_model
    oracles = []
    for e in sub:
        if shapes is not None:
            bm = task.body_shapes[task._env_shape_ids[e]]
        o = PhysOracle(bm, default_params(h=1.0 / (60.0 * substeps), joint_limits=int(limits), solver_type={"pgs": 0, "tgs": 1}[solver],
                                           friction_frame={"world": 0, "velocity": 1}[env.get("friction_frame", "world")]), kp=bm.kp.astype(np.float32), kd=bm.kd.astype(np.float32))
        o.set_state(root[e], dpos[e], dvel[e])
        o.attach_ball(task.racket_geometry)
        oracles.append(o)
    if subset is None:
        task._rigid_body_state[:] = T(np.stack([o.get_state()[3] for o in oracles]).reshape(n * 24, 13))
    else:  # (the launch geometry of the other envs: the rigid-body state the reset left, i.e. the unperturbed reference pose)
        rbv = task._rigid_body_state.view(n, 24, 13)
        rbv[T(sub, torch.long)] = T(np.stack([o.get_state()[3] for o in oracles]))
    ball = _launch(task, rng, mode)
    task._ball_root_states[:] = T(ball)
    hits_total, ground_total, body_total, multi_total = 0, 0, 0, 0
    has_hit = np.zeros(len(sub), dtype=bool)
    for step in range(steps):
        act = np.concatenate([N(task._target_dof_pos) + rng.normal(0, 0.17, (n, 69)), rng.normal(0, 0.17, (n, 6))], axis=1).astype(np.float32)
        rb0 = N(task._rigid_body_state).reshape(n, 24, 13).copy()
        dpos_before = N(task._dof_pos).copy()
        ball_before = N(task._ball_root_states).copy()
        a = T(act)
        task.pre_physics_step(a)
        task._physics_step()
        torch.cuda.synchronize()
        _, pd, _, force, torque = O.pre_physics(act, N(task.reset_buf), dpos_before, rb0[:, 0, 3:7], bm.kp.astype(np.float32))
        per_sim, hit, bc, rbs, ids, cf, bbf, cfs, sens, own, margin = [], [], [], [], [], [], [], [], [], [], []
        # the oracle solves with the hull vertices the kernel selected in every substep (teacher forcing, as in test_gpu_physics.py: selection
        # and solve are judged separately); its own picks must agree except where the selection rule is tied
        ids_sub = N(task.debug_contacts_substeps())[sub]
        assert np.array_equal(ids_sub[:, -1], N(task.debug_contacts())[sub])
        for k, e in enumerate(sub):
            oracles[k].set_ball(ball_before[e])
            # conditioning of the oracle's own step (float32-rounding perturbations of its inputs): see test_gpu_physics.py
            sens.append(oracles[k].ball_sensitivity(pd_target=pd[e], ext_force=force[e], ext_torque=torque[e], nsub=2 * substeps, hold=substeps, sub_per_sim=substeps, seed=step,
                                                    forced_ids=ids_sub[k]))
            c, _, i, ps, h, b = oracles[k].step_ball(pd_target=pd[e], ext_force=force[e], ext_torque=torque[e], nsub=2 * substeps, hold=substeps, sub_per_sim=substeps,
                                                     forced_ids=ids_sub[k])
            per_sim.append(ps); hit.append(h); bc.append(b); rbs.append(oracles[k].get_state()[3]); ids.append(i); cf.append(c); bbf.append(oracles[k].ball_body_force); cfs.append(oracles[k].contact_force_sum)
            own.append(oracles[k].own_ids); margin.append(oracles[k].margins)
        per_sim, hit, bc, rbs, ids, cf, bbf, cfs, own, margin = map(np.stack, (per_sim, hit, bc, rbs, ids, cf, bbf, cfs, own, margin))
        sens = {k: np.stack([s[k] for s in sens]) for k in sens[0]}
        assert np.array_equal(ids, ids_sub[:, -1]), "the forced hull vertices are the ones the oracle used"
        differ = (own != ids_sub).any(axis=-1)  # [n, nsub, 24]
        touching = (own >= 0).any(axis=-1) | (ids_sub >= 0).any(axis=-1)
        if differ.any():
            print("[selection] %s step %d: %d of %d touching (env, substep, body) triples selected differently; largest decision margin among them %.2e m"
                  % (mode, step, int(differ.sum()), int(touching.sum()), float(margin[differ].max())))
            assert differ.sum() <= max(1, 0.01 * touching.sum()) and margin[differ].max() < TIE_TOL, "a selection difference is not a tie of the rule"
        got_ps = N(task._ball_states_per_sim)[sub]
        ball_before = ball_before[sub]
        close(got_ps[..., 0:3], per_sim[..., 0:3], 2e-5, "%s ball pos (step %d)" % (mode, step))
        qs = np.sign(np.sum(got_ps[..., 3:7] * per_sim[..., 3:7], -1, keepdims=True))
        close(got_ps[..., 3:7] * qs, per_sim[..., 3:7], 1e-4, "ball quat", sens=sens["ball"][..., 3:7])  # (integrates the spin: conditioned like it)
        close(got_ps[..., 7:10], per_sim[..., 7:10], 5e-4, "%s ball vel (step %d)" % (mode, step), sens=sens["ball"][..., 7:10])
        close(got_ps[..., 10:13], per_sim[..., 10:13], 5e-4, "%s ball spin (step %d)" % (mode, step), sens=sens["ball"][..., 10:13])
        assert np.array_equal(N(task._ball_root_states)[sub], got_ps[:, -1])
        assert np.array_equal(N(task._racket_ball_contact_per_sim)[sub], hit), "racket hit flags"
        # the reference's sticky flag and its per-step edge (humanoid_smpl_im_mvae.py:773-779), kept by the physics launch itself
        # (only with sim.substeps <= 2, :769)
        now = (hit.any(axis=1) & ~has_hit) if substeps <= 2 else np.zeros(len(sub), dtype=bool)
        has_hit |= now
        assert np.array_equal(N(task._has_racket_ball_contact_now)[sub], now) and np.array_equal(N(task._has_racket_ball_contact)[sub], has_hit)
        close(N(task._ball_contact_forces)[sub], bc, 2e-2, "contact forces on the ball", sens=sens["bc"][:, 0:2])
        close(N(task._ball_body_contact_force)[sub], bbf, 2e-2, "contact force on the ball from the humanoid's links", sens=sens["bc"][:, 2])
        rb = N(task._rigid_body_state).reshape(n, 24, 13)[sub]
        close(rb[..., 0:3], rbs[..., 0:3], 2e-5, "rb pos")
        close(rb[..., 7:13], rbs[..., 7:13], 1e-3, "rb vel", sens=sens["rb"][..., 7:13])
        close(N(task._contact_forces)[sub], cf, 2e-2, "net contact forces (the racket's link carries the reaction of the ball)", sens=sens["cf"])
        close(N(task._contact_forces_sum)[sub], cfs, 2e-2, "_contact_forces_sum: net contact forces summed over the two simulate() calls", sens=sens["cfs"])
        if lift == 0.0:  # standing on the ground: the feet carry the weight in both simulate() calls
            assert np.abs(cfs - cf).max() > 1.0, "the first simulate() call contributes"
        # the racket rigid body = the wrist frame moved by the weld offset
        Rw = Rotation.from_quat(rbs[:, rl, 3:7]).as_matrix()
        off = np.einsum("nij,j->ni", Rw, task.racket_geometry["racket_offset"])
        close(N(task._racket_rb_state)[sub][:, 0:3], rbs[:, rl, 0:3] + off, 2e-5, "racket pos")
        close(N(task._racket_rb_state)[sub][:, 7:10], rbs[:, rl, 7:10] + np.cross(rbs[:, rl, 10:13], off), 1e-3, "racket vel")
        multi_total += sum(int(o.max_hull_points >= 2) for o in oracles)  # envs whose ball touched two links' hulls in one substep
        hits_total += int(hit.sum())
        ground_total += int(((ball_before[:, 9] < -0.5) & (got_ps[:, -1, 9] > 0)).sum())  # balls that bounced within this control step
        body_total += int(((np.linalg.norm(got_ps[:, -1, 7:10] - ball_before[:, 7:10], axis=1) > 3.0) & (hit.sum(1) == 0) & (got_ps[:, -1, 2] > 0.1)).sum())  # deflected by a hull
        task.post_physics_step()
    if mode == "hit":
        # (the per-call flag looks at the call's LAST substep: one in six with sim.substeps 6)
        assert hits_total >= (n // 4 if substeps <= 2 else 2), "the fixture must produce racket hits (%d)" % hits_total
    if mode == "ground":
        assert ground_total >= n // 4, ground_total
    if mode == "joint":
        assert multi_total >= n // 8, "the fixture must produce balls that touch two links at once (%d)" % multi_total
    if mode == "body":
        assert body_total >= n // 4, "the fixture must produce ball x hull hits (%d)" % body_total
    task.close()
    return {"hits": hits_total, "ground": ground_total, "body": body_total, "multi": multi_total}


@pytest.mark.parametrize("solver", ["pgs", "tgs"])
def test_racket_ball_full_size_sample_matches_oracle(mlib, solver):
    """BASELINE config 4 as worded, at its size: 8192 envs with racket + ball, joint limits on, a ball served at every player (bench.py's
    serve, started close enough to arrive within the test); 64 of the envs - first, last and a spread - against their own oracles over
    three control steps: substep jobs, pairing by load and the tail of a full-size launch with the ball kernel."""
    n = 8192
    subset = sorted(set([0, 1, 2, n - 2, n - 1] + list(np.random.default_rng(9).integers(0, n, size=59))))
    got = _ball_step_vs_oracle(mlib, "serve", 0.0, True, "djokovic", n=n, subset=subset, steps=3, solver=solver)
    print("[racket-ball 8192 %s] sampled envs: racket hits %d, ground bounces %d, deflections by a link's hull %d, balls on two hulls at once %d" % (solver, got["hits"], got["ground"], got["body"], got["multi"]))
    assert got["hits"] + got["body"] + got["ground"] >= 8, "the serve must reach something in the sampled envs: %s" % got


def test_bounce_and_hit_flags(mlib):
    """The reference's per-simulate() bookkeeping (humanoid_smpl_im_mvae.py:731-737, 773-779) on top of the engine's outputs."""
    n = 8
    task = make_rb_task(n, mlib)
    task.reset_with_times(None, torch.full((n,), 0.3, device=DEV))
    pos = torch.tensor([[5.0, 5.0, 0.5]], device=DEV).repeat(n, 1)
    task.reset_balls(torch.arange(n), pos, torch.tensor([[1.0, 0.0, -6.0]], device=DEV).repeat(n, 1), torch.zeros((n, 3), device=DEV))
    a = torch.cat([task._target_dof_pos.clone(), torch.zeros((n, 6), device=DEV)], dim=1).contiguous()
    seen = []
    for _ in range(6):
        task.step(a.clone())
        seen.append((task._has_bounce_now.clone(), task._ball_root_states[:, 2].clone()))
    torch.cuda.synchronize()
    assert task._has_bounce.all() and sum(int(s[0].any()) for s in seen) == 1  # flagged once, when the ball first comes within 4 R of the ground
    assert (task._bounce_pos[:, 2] <= 4 * 0.032 + 1e-6).all() and (task._bounce_pos[:, 2] > 0).all()
    assert not task._has_racket_ball_contact.any()
    assert task._ball_root_states[:, 2].min() > 0.0  # the ball did not tunnel
    task.close()


@pytest.mark.parametrize("solver", ["pgs", "tgs"])
@pytest.mark.parametrize("n,timeout_spins", [(3, None), (2048, None), (2048, "0")])
def test_substep_jobs_are_invisible_with_ball(mlib, n, timeout_spins, solver):
    """Racket + ball + joint limits through substep jobs (the ball's state and its aerodynamic force are handed over with the
    humanoid's; the flags are kept through system-scope accesses): bit-identical to one workgroup per env pair, step after step.
    timeout_spins "0": every job whose predecessor is not done at its first look recomputes the earlier substeps itself (the recovery
    path: the replayed substeps must leave the per-call flags and the force accumulator alone)."""
    import warnings

    outs = []
    for jobs in (False, True):
        task = make_rb_task(n, mlib, contact_solver=solver, substep_jobs=2 * int(jobs), debug_contacts=0, job_timeout_spins=-1 if (jobs and timeout_spins is not None) else 0)
        g = torch.Generator(device=DEV)
        g.manual_seed(23)
        task.reset_with_times(None, torch.rand(n, device=DEV, generator=g) * 0.8)
        root = task._humanoid_root_states[:, 0:3]
        jit = torch.rand((n, 3), device=DEV, generator=g)
        # served at the players from 2.5 m: body hits, racket hits and ground bounces all occur within the steps below
        task.reset_balls(torch.arange(n, device=DEV), root + torch.tensor([2.5, 0.0, 0.3], device=DEV) + jit * 0.6,
                         torch.tensor([-20.0, 0.0, 1.0], device=DEV) + (jit - 0.5) * torch.tensor([6.0, 4.0, 4.0], device=DEV),
                         torch.tensor([0.0, -120.0, 0.0], device=DEV).expand(n, 3))
        snaps = []
        for k in range(10):
            a = torch.cat([task._target_dof_pos + 0.4 * torch.randn((n, 69), device=DEV, generator=g), 0.3 * torch.randn((n, 6), device=DEV, generator=g)], dim=1).contiguous()
            task.step(a)
            snaps.append([N(task._rigid_body_state).copy(), N(task._ball_root_states).copy(), N(task._ball_states_per_sim).copy(), N(task._contact_forces).copy(),
                          N(task._ball_contact_forces).copy(), N(task._ball_body_contact_force).copy(), N(task._racket_ball_contact_per_sim).copy(),
                          N(task._has_bounce).copy(), N(task._has_bounce_now).copy(), N(task._bounce_pos).copy(), N(task._has_racket_ball_contact).copy(),
                          N(task.rew_buf).copy(), N(task.reset_buf).copy(), N(task._contact_forces_sum).copy(), N(task._has_racket_ball_contact_now).copy()])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            task.check()
        if jobs and timeout_spins is not None:
            assert task.job_recoveries() > 100, task.job_recoveries()
        outs.append(snaps)
        task.close()
    if n > 100:
        assert any(np.abs(s[5]).max() > 0 for s in outs[0]), "the fixture must produce ball x hull contacts"
        assert outs[0][-1][7].any(), "... and bounces"
    for k, (sa, sb) in enumerate(zip(*outs)):
        for j, (x, y) in enumerate(zip(sa, sb)):
            assert np.array_equal(x, y), "step %d, tensor %d: %d of %d values differ" % (k, j, int((x != y).sum()), x.size)


def test_aero_force_and_bounce_flags_match_reference_vectors(mlib):
    """tests/golden/ball_aero.npz: inputs and outputs of the reference's own apply_external_force_to_ball (drag + Magnus force, bounce flags
    from the ball height at the start of a simulate() call).  The kernel holds the force over a simulate() call, so a ball in free flight
    changes its velocity by 2 h (g + F / m) in the first call of the step."""
    import os

    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ball_aero.npz"))
    st, had, want_f = G["state_a"].copy(), G["has_bounce_in_a"], G["force_a"]
    n = len(st)
    task = make_rb_task(n, mlib, debug_contacts=0)
    assert task.sim_params.substeps == int(G["substeps_a"]) and task.cfg_v2p.get("spin_scale", 1.0) == float(G["spin_scale_a"])
    task.reset_with_times(None, torch.full((n,), 0.3, device=DEV))
    st[:, 0:2] += 40.0  # away from the humanoids (neither force nor flags depend on x, y)
    task._ball_root_states[:] = T(st)
    task._has_bounce[:] = torch.as_tensor(had, device=DEV)
    a = torch.cat([task._target_dof_pos.clone(), torch.zeros((n, 6), device=DEV)], dim=1).contiguous()
    task.step(a)
    torch.cuda.synchronize()
    ps = N(task._ball_states_per_sim)
    h, m, R = 1.0 / 120.0, 0.057, 0.032
    # ---- force, from the balls that stay clear of the ground during the first call
    free = (st[:, 2] > R + 0.06) & (st[:, 2] + 2 * h * st[:, 9] > R + 0.06)
    assert free.sum() > 20
    got_f = m * ((ps[:, 0, 7:10].astype(np.float64) - st[:, 7:10]) / (2 * h) - np.array([0.0, 0.0, -9.81]))
    err = np.abs(got_f - want_f)[free].max()
    assert err < 2e-3 * np.abs(want_f).max(), (err, np.abs(want_f).max())  # (float32 velocities differenced over 1/60 s)
    # ---- flags: the first call of the step sees the recorded states, the second the state after the first
    thr = np.float32(4 * R)
    now1 = G["has_bounce_now_a"]
    assert np.array_equal(now1, ~had & (st[:, 2] <= thr))
    now2 = ~had & ~now1 & (ps[:, 0, 2] <= thr)
    assert np.array_equal(N(task._has_bounce_now), now1 | now2)
    assert np.array_equal(N(task._has_bounce), had | now1 | now2)
    bp = N(task._bounce_pos)
    assert np.array_equal(bp[now1], st[now1, 0:3]) and np.array_equal(bp[now2], ps[now2, 0, 0:3]) and np.all(bp[~(now1 | now2)] == 0)
    task.close()


def test_racket_ball_task_builds_from_the_references_physx_block(mlib, capsys):
    """vid2player's tennis yamls state `solver_type: 1` (TGS: vid2player/cfg/im/tennis_im.yaml:39, embodied_pose/cfg/djokovic_im.yaml:41).
    The task built from that block RUNS TGS - racket-arm limit rows and the ball's rows inside its slices - (until round 5 it overrode to
    PGS with a message): bit-identical to an explicit env.contact_solver = 'tgs', different from 'pgs' on the same inputs."""
    physx = {"num_threads": 4, "solver_type": 1, "num_position_iterations": 4, "num_velocity_iterations": 0, "contact_offset": 0.02, "rest_offset": 0.0,
             "bounce_threshold_velocity": 0.2, "max_depenetration_velocity": 10.0, "default_buffer_size_multiplier": 10.0}
    n = 64
    outs = {}
    for name, kw in (("yaml", dict(sim_overrides={"physx": physx})), ("tgs", dict(contact_solver="tgs")), ("pgs", dict(contact_solver="pgs"))):
        task = make_rb_task(n, mlib, debug_contacts=0, **kw)
        assert task.contact_solver == ("pgs" if name == "pgs" else "tgs")
        if name == "yaml":
            assert task.contact_solver_source == "sim.physx.solver_type"
            assert "running PGS" not in capsys.readouterr().out
        g = torch.Generator(device=DEV)
        g.manual_seed(5)
        task.reset_with_times(None, torch.rand(n, device=DEV, generator=g) * 0.8)
        root = task._humanoid_root_states[:, 0:3]
        task.reset_balls(torch.arange(n, device=DEV), root + torch.tensor([1.2, 0.0, 0.3], device=DEV), torch.tensor([[-20.0, 0.0, 1.0]], device=DEV).repeat(n, 1),
                         torch.tensor([[0.0, -120.0, 0.0]], device=DEV).repeat(n, 1))
        for _ in range(4):
            a = torch.cat([task._target_dof_pos + 0.4 * torch.randn((n, 69), device=DEV, generator=g), 0.3 * torch.randn((n, 6), device=DEV, generator=g)], dim=1).contiguous()
            task.step(a)
        task.check()
        assert torch.isfinite(task.obs_buf).all()
        outs[name] = (N(task._rigid_body_state).copy(), N(task._ball_root_states).copy())
        task.close()
    assert np.array_equal(outs["yaml"][0], outs["tgs"][0]) and np.array_equal(outs["yaml"][1], outs["tgs"][1])
    assert np.abs(outs["yaml"][0] - outs["pgs"][0]).max() > 1e-3, "TGS must be a different solver from PGS on the same inputs"
