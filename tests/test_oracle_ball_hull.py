"""The oracle's point x convex-hull distance (GJK, oracle/phys/v2p_phys_oracle.c `hull_closest`) against a quadratic program solved
by scipy: min |sum_i l_i v_i - c|^2, l >= 0, sum l = 1.  CPU only."""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import phys_oracle as po
from vid2player3d_amd.model import load_baked_model


def qp_closest(verts, c):
    n = len(verts)
    res = minimize(lambda l: np.sum((l @ verts - c) ** 2), np.ones(n) / n, jac=lambda l: 2.0 * verts @ (l @ verts - c), method="SLSQP", bounds=[(0.0, 1.0)] * n,
                   constraints=[{"type": "eq", "fun": lambda l: l.sum() - 1.0, "jac": lambda l: np.ones(n)}], options={"ftol": 1e-16, "maxiter": 2000})
    p = res.x @ verts
    return np.linalg.norm(p - c), p


@pytest.mark.parametrize("seed", range(4))
def test_random_clouds(seed):
    rng = np.random.default_rng(seed)
    verts = rng.normal(size=(40, 3)) * np.array([0.2, 0.1, 0.05])
    for _ in range(15):
        c = rng.normal(size=3) * 0.3
        d, p = po.hull_closest(verts, c)
        dr, pr = qp_closest(verts, c)
        assert abs(d - dr) < 2e-6, (c, d, dr)
        if dr > 1e-5:
            assert np.linalg.norm(p - pr) < 2e-4


def test_baked_hulls():
    m = load_baked_model()
    off = np.asarray(m.hull_offsets)
    hv = np.asarray(m.hull_verts, dtype=np.float64)
    rng = np.random.default_rng(3)
    for b in range(24):
        v = hv[off[b]:off[b + 1]]
        ctr, ext = 0.5 * (v.min(0) + v.max(0)), 0.5 * (v.max(0) - v.min(0))
        for _ in range(3):
            c = ctr + rng.normal(size=3) * (ext + 0.05)
            d, p = po.hull_closest(v, c)
            dr, pr = qp_closest(v, c)
            assert abs(d - dr) < 2e-6, (b, c, d, dr)


def test_simple_shapes():
    cube = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=float)
    d, p = po.hull_closest(cube, [3.0, 0.0, 0.0])       # face
    assert abs(d - 2.0) < 1e-12 and np.allclose(p, [1, 0, 0], atol=1e-12)
    d, p = po.hull_closest(cube, [2.0, 2.0, 0.0])       # edge
    assert abs(d - np.sqrt(2.0)) < 1e-12 and np.allclose(p, [1, 1, 0], atol=1e-12)
    d, p = po.hull_closest(cube, [2.0, 2.0, 2.0])       # vertex
    assert abs(d - np.sqrt(3.0)) < 1e-12 and np.allclose(p, [1, 1, 1], atol=1e-12)
    d, p = po.hull_closest(cube, [0.2, -0.3, 0.1])      # inside
    assert d == 0.0 and np.allclose(p, [0.2, -0.3, 0.1])
    d, p = po.hull_closest(cube[:1], [0.0, 0.0, 0.0])   # one vertex
    assert abs(d - np.sqrt(3.0)) < 1e-12
