"""The stand-in for `isaacgym.torch_utils` that the golden generator imports the reference through (oracle/ref_shim: Isaac Gym itself is a
closed binary and absent) is pinned to two independent implementations: scipy's Rotation, and - in the build container, where
/root/reference exists - the reference's OWN quaternion library (poselib/poselib/core/rotation3d.py), which uses the same xyzw convention.
A wrong sign or order in the shim would otherwise be baked into goldens, oracle and kernels alike."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("shim_torch_utils", os.path.join(HERE, "..", "oracle", "ref_shim", "isaacgym", "torch_utils.py"))
S = importlib.util.module_from_spec(spec)
spec.loader.exec_module(S)

RNG = np.random.default_rng(12)


def _rand_quats(n):
    q = RNG.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def _same_rotation(a, b, tol=1e-6):
    d = np.abs(np.sum(np.asarray(a) * np.asarray(b), axis=-1))
    assert np.all(d > 1 - tol), float(d.min())


def test_against_scipy():
    a, b = _rand_quats(200), _rand_quats(200)
    ab = S.quat_mul(torch.tensor(a), torch.tensor(b)).numpy()
    _same_rotation(ab, (Rotation.from_quat(a) * Rotation.from_quat(b)).as_quat())  # Hamilton product, xyzw: R(ab) = R(a) R(b)
    assert np.allclose(S.quat_conjugate(torch.tensor(a)).numpy(), a * [-1, -1, -1, 1])
    ang, ax = RNG.uniform(-6, 6, size=200), RNG.normal(size=(200, 3))
    q = S.quat_from_angle_axis(torch.tensor(ang), torch.tensor(ax)).numpy()
    _same_rotation(q, Rotation.from_rotvec(ang[:, None] * ax / np.linalg.norm(ax, axis=1, keepdims=True)).as_quat())
    x = RNG.uniform(-20, 20, size=500)
    w = S.normalize_angle(torch.tensor(x)).numpy()
    assert np.all(w <= np.pi + 1e-12) and np.all(w >= -np.pi - 1e-12) and np.allclose(np.cos(w), np.cos(x)) and np.allclose(np.sin(w), np.sin(x))
    r, p, y = (RNG.uniform(-3, 3, size=100) for _ in range(3))
    q = S.quat_from_euler_xyz(torch.tensor(r), torch.tensor(p), torch.tensor(y)).numpy()
    _same_rotation(q, Rotation.from_euler("xyz", np.stack([r, p, y], 1)).as_quat())  # extrinsic x, y, z = Rz(yaw) Ry(pitch) Rx(roll)
    assert S.get_axis_params(0.89, 2) == [0.0, 0.0, 0.89] and S.get_axis_params(1.0, 1, x_value=0.5) == [0.5, 1.0, 0.0]
    assert np.allclose(np.linalg.norm(S.quat_unit(torch.tensor(3.0 * a)).numpy(), axis=1), 1.0)


@pytest.mark.skipif(not os.path.exists("/root/reference/poselib/poselib/core/rotation3d.py"), reason="the reference tree is only present in the build container")
def test_against_the_references_own_quaternion_library():
    prev, sys.dont_write_bytecode = sys.dont_write_bytecode, True  # never leave __pycache__ in the read-only reference mount
    sys.path.insert(0, "/root/reference/poselib")
    try:
        from poselib.core import rotation3d as R3
    finally:
        sys.path.pop(0)
        sys.dont_write_bytecode = prev
    a, b = torch.tensor(_rand_quats(100), dtype=torch.float32), torch.tensor(_rand_quats(100), dtype=torch.float32)
    assert torch.allclose(S.quat_mul(a, b), R3.quat_mul(a, b), atol=1e-6)
    assert torch.allclose(S.quat_conjugate(a), R3.quat_conjugate(a))
    ang = torch.tensor(RNG.uniform(-3, 3, size=100), dtype=torch.float32)
    ax = torch.tensor(RNG.normal(size=(100, 3)), dtype=torch.float32)
    _same_rotation(S.quat_from_angle_axis(ang, ax).numpy(), R3.quat_from_angle_axis(ang, ax).numpy(), tol=1e-5)
    # the reference's own rotation test (poselib/core/tests/test_rotation.py:27-32): rotating by q then by its inverse is the identity
    v = torch.tensor(RNG.normal(size=(100, 3)), dtype=torch.float32)
    back = R3.quat_rotate(S.quat_conjugate(a), R3.quat_rotate(a, v))
    assert torch.allclose(back, v, atol=1e-5)
