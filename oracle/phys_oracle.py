"""ctypes front-end of the C physics oracle (oracle/phys/v2p_phys_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "phys", "libv2p_phys_oracle.so")
LIB_FAST = os.path.join(HERE, "phys", "libv2p_phys_oracle_fast.so")   # same source, -O3 -mavx2 -mfma: bench.py's cpu_baseline only
FAST_FLAGS = "-O3 -mavx2 -mfma -ffp-contract=fast -funroll-loops -fopenmp"
NB, NJ, ND = 24, 23, 75


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "phys")])


class OModel(C.Structure):
    _fields_ = [
        ("parents", C.c_int * NB), ("local_pos", C.c_double * (NB * 3)), ("mass", C.c_double * NB), ("com", C.c_double * (NB * 3)),
        ("inertia", C.c_double * (NB * 9)), ("kp", C.c_double * (3 * NJ)), ("kd", C.c_double * (3 * NJ)), ("armature", C.c_double * (3 * NJ)),
        ("hull_offsets", C.c_int * (NB + 1)), ("hull_verts", C.POINTER(C.c_double)),
        ("limit_lo", C.c_double * (3 * NJ)), ("limit_hi", C.c_double * (3 * NJ)),
    ]


class OParams(C.Structure):
    _fields_ = [("h", C.c_double), ("gravity_z", C.c_double), ("mu", C.c_double), ("contact_offset", C.c_double), ("max_depen_vel", C.c_double),
                ("ang_damp", C.c_double), ("max_ang_vel", C.c_double), ("erp", C.c_double), ("n_iter", C.c_int), ("enable_contact", C.c_int), ("solver_type", C.c_int),
                ("joint_limits", C.c_int), ("limit_margin", C.c_double), ("rest_offset", C.c_double), ("friction_frame", C.c_int)]


class OCyl(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("axis", C.c_double * 3), ("half_len", C.c_double), ("radius", C.c_double)]


class OBallParams(C.Structure):
    _fields_ = [("radius", C.c_double), ("mass", C.c_double), ("inertia", C.c_double), ("rest_ground", C.c_double), ("fric_ground", C.c_double),
                ("rest_racket", C.c_double), ("fric_racket", C.c_double), ("rest_body", C.c_double), ("fric_body", C.c_double),
                ("bounce_threshold", C.c_double), ("ang_damp", C.c_double),
                ("max_ang_vel", C.c_double), ("racket_link", C.c_int), ("ncyl", C.c_int), ("body_contacts", C.c_int), ("cyl", OCyl * 2)]


class OBall(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("quat", C.c_double * 4), ("vel", C.c_double * 3), ("angvel", C.c_double * 3)]


class OState(C.Structure):
    _fields_ = [("root_pos", C.c_double * 3), ("root_quat", C.c_double * 4), ("jquat", C.c_double * (NJ * 4)), ("vel", C.c_double * ND)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        assert _lib.v2p_oracle_sizeof_model() == C.sizeof(OModel)
        assert _lib.v2p_oracle_sizeof_state() == C.sizeof(OState)
        assert _lib.v2p_oracle_sizeof_params() == C.sizeof(OParams)
        assert _lib.v2p_oracle_sizeof_ball_params() == C.sizeof(OBallParams) and _lib.v2p_oracle_sizeof_ball() == C.sizeof(OBall)
    return _lib


_lib_fast = None


def lib_fast():
    """The same C source built with FAST_FLAGS (oracle/phys/Makefile): the cpu_baseline leg of bench.py times this one; no test's pass
    criterion uses it."""
    global _lib_fast
    if _lib_fast is None:
        if not os.path.exists(LIB_FAST):
            build()
        _lib_fast = C.CDLL(LIB_FAST)
        assert _lib_fast.v2p_oracle_sizeof_state() == C.sizeof(OState) and _lib_fast.v2p_oracle_sizeof_params() == C.sizeof(OParams)
    return _lib_fast


def default_params(h=1.0 / 120.0, enable_contact=True, **kw):
    """amass_im.yaml:37-52 + humanoid_smpl_im.py:273-276."""
    p = OParams(h=h, gravity_z=-9.81, mu=1.0, contact_offset=0.02, max_depen_vel=10.0, ang_damp=0.01, max_ang_vel=100.0, erp=0.2, n_iter=4,
                enable_contact=int(enable_contact), solver_type=0, joint_limits=0, limit_margin=0.05, rest_offset=0.0, friction_frame=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _omodel(body_model, kp=None, kd=None, armature=None):
    """(OModel, keep-alive array) for a vid2player3d_amd.model.BodyModel."""
    m = OModel()
    m.parents[:] = [int(x) for x in body_model.parents]
    m.local_pos[:] = body_model.local_pos.reshape(-1).tolist()
    m.mass[:] = body_model.mass.tolist()
    m.com[:] = body_model.com.reshape(-1).tolist()
    m.inertia[:] = body_model.inertia.reshape(-1).tolist()
    m.kp[:] = np.asarray(body_model.kp if kp is None else kp, dtype=np.float64).tolist()
    m.kd[:] = np.asarray(body_model.kd if kd is None else kd, dtype=np.float64).tolist()
    m.armature[:] = np.asarray(body_model.armature if armature is None else armature, dtype=np.float64).tolist()
    m.hull_offsets[:] = [int(x) for x in body_model.hull_offsets]
    m.limit_lo[:] = np.asarray(body_model.limit_lower, dtype=np.float64).tolist()
    m.limit_hi[:] = np.asarray(body_model.limit_upper, dtype=np.float64).tolist()
    hv = np.ascontiguousarray(body_model.hull_verts, dtype=np.float64)
    m.hull_verts = _dptr(hv)
    return m, hv


def hull_closest(verts, c):
    """(distance, closest point) of the convex hull of `verts` [n,3] to the point c (the GJK the ball x hull contacts use)."""
    v = np.ascontiguousarray(verts, dtype=np.float64)
    cc = np.ascontiguousarray(c, dtype=np.float64)
    p = np.zeros(3)
    fn = lib().v2p_oracle_hull_closest
    fn.restype = C.c_double
    d = fn(_dptr(v), int(len(v)), _dptr(cc), _dptr(p))
    return float(d), p


def ball_aero(state13, spin_scale=1.0):
    """Aerodynamic force on the ball (drag + Magnus lift) for a ball root state [13], as the oracle evaluates it before a simulate() call."""
    b = OBall()
    r = np.asarray(state13, dtype=np.float64)
    b.pos[:], b.quat[:], b.vel[:], b.angvel[:] = r[0:3].tolist(), r[3:7].tolist(), r[7:10].tolist(), r[10:13].tolist()
    f = np.zeros(3)
    lib().v2p_oracle_ball_aero(C.byref(b), C.c_double(spin_scale), _dptr(f))
    return f


class PhysOracle:
    """One humanoid per instance; batches are Python loops (small cases only)."""

    def __init__(self, body_model, params=None, kp=None, kd=None, armature=None):
        self.lib = lib()
        self.model, self._hv = _omodel(body_model, kp, kd, armature)
        self.params = params or default_params()
        self.state = OState()

    def set_state(self, root13, dof_pos, dof_vel):
        r = np.ascontiguousarray(root13, dtype=np.float64)
        p = np.ascontiguousarray(dof_pos, dtype=np.float64)
        v = np.ascontiguousarray(dof_vel, dtype=np.float64)
        self.lib.v2p_oracle_set_state(C.byref(self.state), _dptr(r), _dptr(p), _dptr(v))

    def get_state(self):
        root = np.zeros(13)
        dp = np.zeros(69)
        dv = np.zeros(69)
        rb = np.zeros((NB, 13))
        self.lib.v2p_oracle_get_state(C.byref(self.model), C.byref(self.state), _dptr(root), _dptr(dp), _dptr(dv), _dptr(rb))
        return root, dp, dv, rb

    def step(self, pd_target=None, ext_force=None, ext_torque=None, nsub=4, hold=2, forced_ids=None, want_selection=False):
        """forced_ids [nsub,24,4] int32: contact vertices to use instead of the selection rule.  want_selection: also return the
        rule's own picks [nsub,24,4] in the state of every substep and their decision margins [nsub,24] (metres)."""
        cf = np.zeros((NB, 3))
        df = np.zeros(69)
        ids = np.full(NB * 4, -1, dtype=np.int32)
        tar = None if pd_target is None else np.ascontiguousarray(pd_target, dtype=np.float64)
        f = None if ext_force is None else np.ascontiguousarray(ext_force, dtype=np.float64)
        t = None if ext_torque is None else np.ascontiguousarray(ext_torque, dtype=np.float64)
        forced = None if forced_ids is None else np.ascontiguousarray(forced_ids, dtype=np.int32).reshape(nsub, NB * 4)
        own = np.full((nsub, NB, 4), -1, dtype=np.int32) if want_selection else None
        mg = np.zeros((nsub, NB)) if want_selection else None
        rc = self.lib.v2p_oracle_step_io(C.byref(self.model), C.byref(self.params), C.byref(self.state), _dptr(tar), _dptr(f), _dptr(t),
                                         int(nsub), int(hold), _dptr(cf), _dptr(df), _iptr(ids), _iptr(forced), _iptr(own), _dptr(mg), None)
        if rc:
            raise RuntimeError("oracle substep failed (mass matrix not positive definite)")
        if want_selection:
            return cf, df, ids.reshape(NB, 4), own, mg
        return cf, df, ids.reshape(NB, 4)

    # ---- racket + ball
    def attach_ball(self, geom, ball=None, material=None, spin_scale=1.0, body_contacts=True):
        """geom: the dict vid2player3d_amd.racket.with_racket returns (or None: ball without racket); ball / material: overrides of
        racket.BALL / racket.BALL_MATERIAL; body_contacts: ball x hull contacts on."""
        from vid2player3d_amd import racket as R

        b, mat = dict(R.BALL, **(ball or {})), dict(R.BALL_MATERIAL, **(material or {}))
        bp = OBallParams(radius=b["radius"], mass=b["mass"], inertia=b["inertia"], racket_link=-1, ncyl=0, body_contacts=int(body_contacts), **mat)
        if geom is not None:
            bp.racket_link = int(geom["racket_link"])
            bp.ncyl = len(geom["cylinders"])
            for k, c in enumerate(geom["cylinders"]):
                bp.cyl[k].center[:] = list(c["center"]); bp.cyl[k].axis[:] = list(c["axis"])
                bp.cyl[k].half_len, bp.cyl[k].radius = float(c["half_len"]), float(c["radius"])
        self.ball_params, self.ball, self.spin_scale = bp, OBall(), float(spin_scale)
        self.ball.quat[:] = [0, 0, 0, 1]

    def set_ball(self, root13):
        r = np.asarray(root13, dtype=np.float64)
        self.ball.pos[:], self.ball.quat[:], self.ball.vel[:], self.ball.angvel[:] = r[0:3].tolist(), r[3:7].tolist(), r[7:10].tolist(), r[10:13].tolist()

    def get_ball(self):
        return np.array(list(self.ball.pos) + list(self.ball.quat) + list(self.ball.vel) + list(self.ball.angvel))

    def step_ball(self, pd_target=None, ext_force=None, ext_torque=None, nsub=4, hold=2, sub_per_sim=2, forced_ids=None):
        """One control step with the ball: returns (contact_force [24,3], dof_force, contact ids, ball state after each simulate()
        [nsim,13], racket-hit flag per simulate() [nsim], force on the ball from racket / ground in the last substep [2,3]).
        forced_ids [nsub,24,4] int32: hull vertices to use instead of the selection rule (teacher forcing, like step()); the rule's own
        picks and their decision margins are left in self.own_ids [nsub,24,4] / self.margins [nsub,24]."""
        cf, df, ids = np.zeros((NB, 3)), np.zeros(69), np.full(NB * 4, -1, dtype=np.int32)
        nsim = nsub // sub_per_sim
        per_sim, hit, bc, cfs = np.zeros((nsim, 13)), np.zeros(nsim, dtype=np.int32), np.zeros(9), np.zeros((NB, 3))
        tar = None if pd_target is None else np.ascontiguousarray(pd_target, dtype=np.float64)
        f = None if ext_force is None else np.ascontiguousarray(ext_force, dtype=np.float64)
        t = None if ext_torque is None else np.ascontiguousarray(ext_torque, dtype=np.float64)
        nh = C.c_int(0)
        forced = None if forced_ids is None else np.ascontiguousarray(forced_ids, dtype=np.int32).reshape(nsub, NB * 4)
        self.own_ids, self.margins = np.full((nsub, NB, 4), -1, dtype=np.int32), np.zeros((nsub, NB))
        rc = self.lib.v2p_oracle_step_ball_io(C.byref(self.model), C.byref(self.params), C.byref(self.state), _dptr(tar), _dptr(f), _dptr(t), int(nsub), int(hold),
                                              int(sub_per_sim), _dptr(cf), _dptr(df), _iptr(ids), C.byref(self.ball_params), C.byref(self.ball),
                                              C.c_double(self.spin_scale), _dptr(per_sim), _iptr(hit), _dptr(bc), C.byref(nh), _dptr(cfs),
                                              _iptr(forced), _iptr(self.own_ids), _dptr(self.margins))
        self.max_hull_points = int(nh.value)  # most ball x hull points active in one substep of the step
        self.contact_force_sum = cfs  # net contact forces of the links summed over the simulate() calls of the step
        if rc:
            raise RuntimeError("oracle ball step failed (%d)" % rc)
        self.ball_body_force = bc[6:9].copy()  # force on the ball from the humanoid's links, last substep
        return cf, df, ids.reshape(NB, 4), per_sim, hit, bc[:6].reshape(2, 3)

    def ball_sensitivity(self, pd_target=None, ext_force=None, ext_torque=None, nsub=4, hold=2, sub_per_sim=2, trials=16, eps_pos=2e-7, eps_vel=1e-6, seed=0,
                         forced_ids=None):
        """CONDITIONING of the control step with the ball this instance is about to take (call it before step_ball(); humanoid and ball
        states are left untouched) - BatchOracle.sensitivity for one env with a ball: the largest change, over `trials` runs whose
        humanoid and ball states are perturbed at float32-rounding size, of the rigid-body state [24,13], the net contact forces of the
        last substep / summed over the simulate() calls [24,3], the ball after each simulate() [nsim,13] and the forces on the ball [3,3]."""
        nst, nbl = C.sizeof(OState), C.sizeof(OBall)
        s0, b0 = C.string_at(C.addressof(self.state), nst), C.string_at(C.addressof(self.ball), nbl)
        st = np.frombuffer((C.c_char * nst).from_address(C.addressof(self.state)), dtype=np.float64)  # pos 3 | quats 96 | velocity 75
        rng = np.random.default_rng(seed)

        def run():
            cf, _, _, ps, _, bc = self.step_ball(pd_target, ext_force, ext_torque, nsub, hold, sub_per_sim, forced_ids=forced_ids)
            return {"rb": self.get_state()[3], "cf": cf.copy(), "cfs": self.contact_force_sum.copy(), "ball": ps.copy(),
                    "bc": np.concatenate([bc.reshape(-1), self.ball_body_force]).reshape(3, 3)}

        base = run()
        sens = {k: np.zeros_like(v) for k, v in base.items()}
        for _ in range(trials):
            C.memmove(C.addressof(self.state), s0, nst)
            C.memmove(C.addressof(self.ball), b0, nbl)
            st[0:3] += eps_pos * rng.normal(size=3)
            q = st[3:99].reshape(24, 4)
            q += eps_pos * rng.normal(size=q.shape)
            q /= np.linalg.norm(q, axis=-1, keepdims=True)
            st[99:] += eps_vel * rng.normal(size=st.size - 99)
            b = self.get_ball()
            b[0:3] += eps_pos * rng.normal(size=3)
            b[7:13] += eps_vel * rng.normal(size=6) * np.array([1, 1, 1, 30, 30, 30])  # (spins are tens of rad/s: one float32 ulp is larger there)
            self.set_ball(b)
            out = run()
            for k in sens:
                np.maximum(sens[k], np.abs(out[k] - base[k]), out=sens[k])
        C.memmove(C.addressof(self.state), s0, nst)
        C.memmove(C.addressof(self.ball), b0, nbl)
        return sens

    def diagnostics(self):
        out = np.zeros(8)
        self.lib.v2p_oracle_diagnostics(C.byref(self.model), C.byref(self.params), C.byref(self.state), _dptr(out))
        return {"ke": out[0], "pe": out[1], "P": out[2:5].copy(), "L": out[5:8].copy()}


class BatchOracle:
    """n independent humanoids stepped by one C call (OpenMP over envs).  `models` is one BodyModel or a list of them with
    `model_of` [n] naming each env's; gains default to the models' own."""

    def __init__(self, models, n, params=None, model_of=None, gains_f32=True, threads=0, fast=False):
        self.lib = lib_fast() if fast else lib()
        models = list(models) if isinstance(models, (list, tuple)) else [models]
        cast = (lambda x: x.astype(np.float32)) if gains_f32 else (lambda x: x)
        self._m = [_omodel(bm, cast(bm.kp), cast(bm.kd)) for bm in models]
        self._mptr = (C.POINTER(OModel) * len(models))(*[C.pointer(m) for m, _ in self._m])
        self.model_of = None if model_of is None else np.ascontiguousarray(model_of, dtype=np.int32)
        self.params = params or default_params()
        self.n = int(n)
        self.states = (OState * self.n)()
        self.threads = int(threads)

    def set_state(self, root13, dof_pos, dof_vel):
        r = np.ascontiguousarray(root13, dtype=np.float64)
        p = np.ascontiguousarray(dof_pos, dtype=np.float64)
        v = np.ascontiguousarray(dof_vel, dtype=np.float64)
        for e in range(self.n):
            self.lib.v2p_oracle_set_state(C.byref(self.states[e]), _dptr(r[e]), _dptr(p[e]), _dptr(v[e]))

    def _state_view(self):
        """The env states as a float64 array [n, 174]: root pos 3 | root quat 4 | 23 joint quats | generalised velocity 75."""
        return np.frombuffer(self.states, dtype=np.float64).reshape(self.n, C.sizeof(OState) // 8)

    def sensitivity(self, pd_target, ext_force, ext_torque, nsub=4, hold=2, forced_ids=None, trials=32, eps_pos=2e-7, eps_vel=1e-6, seed=0):
        """CONDITIONING of the step this batch is about to take (call it before step(); the states are left untouched): the step is
        run from the current states and from `trials` copies whose inputs are perturbed at the level of float32 rounding - positions and
        quaternion components by eps_pos N(0,1) (2e-7: about one ulp of a coordinate of 1 .. 2 m), velocities by eps_vel N(0,1) - and
        the largest change of every output element over the trials is returned (same keys and shapes as step()).  The step map of this
        model is piecewise linear but not contractive: box friction bounded by the CURRENT normal impulse couples the rows
        non-symmetrically, and in rare states one substep multiplies a velocity perturbation by 10^2 and more (tools/gain_probe.py).
        A float32 evaluation can be no closer to the float64 result than this; the parity tests add a multiple of it to their
        per-element bounds instead of allowing a share of the envs to miss them.  (trials: the largest change over a handful of random
        directions is itself a noisy estimate of the gain - with 8 trials one env in 16384 sat at 24 x its estimate, at 1.6 x with another
        draw and at 3.5 x with 64 trials, profiles/r04f_sensitivity_trials.txt; 32 keeps the estimate within the tests' factor.)"""
        st = self._state_view()
        saved = st.copy()
        rng = np.random.default_rng(seed)
        keys = ("root", "dpos", "dvel", "rb", "cf", "df")
        # (the trials run through the -O3 build of the same source: an estimate of a gain needs no bit-exact arithmetic, and the parity
        # tests spend most of their time here.  Switches set through v2p_oracle_experiment live in the -O2 library only.)
        lib0 = self.lib
        try:
            self.lib = lib_fast()
        except Exception:
            pass
        try:
            return self._sensitivity(st, saved, rng, keys, pd_target, ext_force, ext_torque, nsub, hold, forced_ids, trials, eps_pos, eps_vel)
        finally:
            self.lib = lib0
            st[:] = saved

    def _sensitivity(self, st, saved, rng, keys, pd_target, ext_force, ext_torque, nsub, hold, forced_ids, trials, eps_pos, eps_vel):
        base = self.step(pd_target, ext_force, ext_torque, nsub, hold, forced_ids)
        sens = {k: np.zeros_like(base[k]) for k in keys}
        for _ in range(trials):
            st[:] = saved
            st[:, 0:3] += eps_pos * rng.normal(size=(self.n, 3))
            q = st[:, 3:99].reshape(self.n, 24, 4)
            q += eps_pos * rng.normal(size=q.shape)
            q /= np.linalg.norm(q, axis=-1, keepdims=True)
            st[:, 99:] += eps_vel * rng.normal(size=(self.n, ND))
            out = self.step(pd_target, ext_force, ext_torque, nsub, hold, forced_ids)
            for k in keys:
                np.maximum(sens[k], np.abs(out[k] - base[k]), out=sens[k])
        st[:] = saved
        return sens

    def step(self, pd_target, ext_force, ext_torque, nsub=4, hold=2, forced_ids=None, want_selection=False):
        n = self.n
        out = {"cf": np.zeros((n, NB, 3)), "df": np.zeros((n, 69)), "ids": np.full((n, NB, 4), -1, dtype=np.int32), "root": np.zeros((n, 13)),
               "dpos": np.zeros((n, 69)), "dvel": np.zeros((n, 69)), "rb": np.zeros((n, NB, 13))}
        tar = np.ascontiguousarray(pd_target, dtype=np.float64)
        f = np.ascontiguousarray(ext_force, dtype=np.float64)
        t = np.ascontiguousarray(ext_torque, dtype=np.float64)
        forced = None if forced_ids is None else np.ascontiguousarray(forced_ids, dtype=np.int32).reshape(n, nsub, NB * 4)
        if want_selection:
            out["own"] = np.full((n, nsub, NB, 4), -1, dtype=np.int32)
            out["margin"] = np.zeros((n, nsub, NB))
        # how close any row update came to the other side of its clamp, as a change of the row's relative velocity (min over the step)
        out["clamp"] = np.full(n, 1e30)
        failed = self.lib.v2p_oracle_step_batch(self._mptr, _iptr(self.model_of), C.byref(self.params), self.states, n, _dptr(tar), _dptr(f), _dptr(t),
                                                int(nsub), int(hold), _dptr(out["cf"]), _dptr(out["df"]), _iptr(out["ids"]), _iptr(forced),
                                                _iptr(out.get("own")), _dptr(out.get("margin")), _dptr(out["root"]), _dptr(out["dpos"]),
                                                _dptr(out["dvel"]), _dptr(out["rb"]), self.threads, _dptr(out["clamp"]))
        if failed:
            raise RuntimeError("oracle step failed in %d envs (mass matrix not positive definite)" % failed)
        return out
