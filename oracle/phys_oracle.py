"""ctypes front-end of the C physics oracle (oracle/phys/v2p_phys_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "phys", "libv2p_phys_oracle.so")
NB, NJ, ND = 24, 23, 75


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "phys")])


class OModel(C.Structure):
    _fields_ = [
        ("parents", C.c_int * NB), ("local_pos", C.c_double * (NB * 3)), ("mass", C.c_double * NB), ("com", C.c_double * (NB * 3)),
        ("inertia", C.c_double * (NB * 9)), ("kp", C.c_double * (3 * NJ)), ("kd", C.c_double * (3 * NJ)), ("armature", C.c_double * (3 * NJ)),
        ("hull_offsets", C.c_int * (NB + 1)), ("hull_verts", C.POINTER(C.c_double)),
    ]


class OParams(C.Structure):
    _fields_ = [("h", C.c_double), ("gravity_z", C.c_double), ("mu", C.c_double), ("contact_offset", C.c_double), ("max_depen_vel", C.c_double),
                ("ang_damp", C.c_double), ("max_ang_vel", C.c_double), ("erp", C.c_double), ("n_iter", C.c_int), ("enable_contact", C.c_int)]


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
    return _lib


def default_params(h=1.0 / 120.0, enable_contact=True, **kw):
    """amass_im.yaml:37-52 + humanoid_smpl_im.py:273-276."""
    p = OParams(h=h, gravity_z=-9.81, mu=1.0, contact_offset=0.02, max_depen_vel=10.0, ang_damp=0.01, max_ang_vel=100.0, erp=0.2, n_iter=4,
                enable_contact=int(enable_contact))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PhysOracle:
    """One humanoid per instance; batches are Python loops (small cases only)."""

    def __init__(self, body_model, params=None, kp=None, kd=None, armature=None):
        self.lib = lib()
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
        self._hv = np.ascontiguousarray(body_model.hull_verts, dtype=np.float64)
        m.hull_verts = _dptr(self._hv)
        self.model = m
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

    def step(self, pd_target=None, ext_force=None, ext_torque=None, nsub=4, hold=2):
        cf = np.zeros((NB, 3))
        df = np.zeros(69)
        ids = np.full(NB * 4, -1, dtype=np.int32)
        tar = None if pd_target is None else np.ascontiguousarray(pd_target, dtype=np.float64)
        f = None if ext_force is None else np.ascontiguousarray(ext_force, dtype=np.float64)
        t = None if ext_torque is None else np.ascontiguousarray(ext_torque, dtype=np.float64)
        rc = self.lib.v2p_oracle_step(C.byref(self.model), C.byref(self.params), C.byref(self.state), None if tar is None else _dptr(tar),
                                      None if f is None else _dptr(f), None if t is None else _dptr(t), int(nsub), int(hold), _dptr(cf),
                                      _dptr(df), ids.ctypes.data_as(C.POINTER(C.c_int)))
        if rc:
            raise RuntimeError("oracle substep failed (mass matrix not positive definite)")
        return cf, df, ids.reshape(NB, 4)

    def diagnostics(self):
        out = np.zeros(8)
        self.lib.v2p_oracle_diagnostics(C.byref(self.model), C.byref(self.params), C.byref(self.state), _dptr(out))
        return {"ke": out[0], "pe": out[1], "P": out[2:5].copy(), "L": out[5:8].copy()}
