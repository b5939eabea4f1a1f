"""Per-clip body assets compiled on the device (csrc/shape_compile.hip, v2p_shapes_compile): one wavefront per (shape, body)
builds the convex hull of the body's vertex cloud, reduces it to <= 64 support vertices and integrates the mass properties of their hull.
The checker is the numpy statement of the same algorithm (body_shapes.body_from_clouds, itself checked against scipy's qhull in
tests/test_body_shapes.py)."""
import time

import numpy as np
import pytest
import torch

from vid2player3d_amd import body_shapes as bs
from vid2player3d_amd.model import load_baked_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _same_body(got, want, what):
    assert np.array_equal(got.hull_offsets, want.hull_offsets), what + ": hull vertex counts"
    assert np.array_equal(got.hull_verts, want.hull_verts), what + ": hull vertices (the same points of the cloud, bit for bit)"
    assert np.allclose(got.local_pos, want.local_pos, rtol=0, atol=1e-15)
    assert np.abs(got.mass / want.mass - 1.0).max() < 1e-9, what
    assert np.abs(got.com - want.com).max() < 1e-9, what
    assert np.abs(got.inertia - want.inertia).max() < 1e-9 * np.abs(want.inertia).max(), what
    assert np.allclose(got.kp, want.kp, rtol=1e-9) and np.allclose(got.kd, want.kd, rtol=1e-9)  # the gains follow the total mass


def test_device_compiler_equals_the_numpy_statement_on_a_shape_family():
    base = load_baked_model()
    clouds, rest = bs.deform_clouds(base, bs.family_params(8, seed=1))
    got = bs.bodies_from_clouds_device(base, clouds, rest, DEV)
    for s in range(8):
        want = bs.body_from_clouds(base, [c[s] for c in clouds], rest[s])
        _same_body(got[s], want, "shape %d" % s)
    # and through the public entry (what bench.py --per-clip-shapes and the tests call)
    fam = bs.synthetic_shape_family(base, 8, seed=1, device=DEV)
    for a, b in zip(fam, got):
        assert np.array_equal(a.hull_verts, b.hull_verts) and np.array_equal(a.mass, b.mass)


def test_identity_and_small_clouds_keep_every_hull_vertex():
    """Clouds of at most 64 hull vertices are not reduced: the baked body's own hulls come back (test_body_shapes' identity test, on the device)."""
    base = load_baked_model()
    clouds, rest = bs.clouds_of(base)
    got = bs.bodies_from_clouds_device(base, [c[None] for c in clouds], rest[None], DEV)[0]
    assert np.array_equal(got.hull_offsets, base.hull_offsets) and np.allclose(got.local_pos, base.local_pos)
    assert abs(got.total_mass / base.total_mass - 1.0) < 1e-6
    assert np.allclose(got.com, base.com, atol=1e-6) and np.allclose(got.inertia, base.inertia, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("seed,n,aniso", [(0, 40, (1, 1, 1)), (1, 300, (1.0, 0.3, 2.0)), (2, 1500, (0.2, 0.2, 1.0)), (3, 8, (1, 1, 1)), (4, 4, (1, 1, 1))])
def test_random_clouds_against_numpy_hull_and_reduction(seed, n, aniso):
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, 3)) * np.array(aniso) + rng.normal(size=3)
    r = bs.compile_clouds_device(pts, [0, n], DEV, density=900.0)
    keep = bs.reduce_hull(pts, 64)
    k = int(r["num_verts"][0])
    assert k == len(keep) and np.array_equal(r["vert_ids"][0, :k], keep)
    assert np.array_equal(r["verts"][0, :k], pts[keep])
    _, faces = bs.convex_hull(pts[keep])
    m, com, inertia = bs.hull_mass_properties_faces(pts[keep], faces, 900.0)
    assert abs(r["mass"][0] / m - 1.0) < 1e-10 and np.abs(r["com"][0] - com).max() < 1e-10 * max(1.0, np.abs(com).max())
    assert np.abs(r["inertia"][0] - inertia).max() < 1e-9 * np.abs(inertia).max()


def test_many_ragged_jobs_in_one_launch():
    """Jobs of very different sizes (4 ... 900 points) in one launch, more jobs than persistent waves: each equals its own numpy result."""
    rng = np.random.default_rng(11)
    sizes = rng.integers(4, 900, size=700)
    sizes[:3] = (4, 5, 899)
    clouds = [rng.normal(size=(int(k), 3)) * rng.uniform(0.05, 0.3, size=3) for k in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)])
    r = bs.compile_clouds_device(np.concatenate(clouds), off, DEV)
    for j in list(range(0, 700, 37)) + [0, 1, 2, 699]:
        keep = bs.reduce_hull(clouds[j], 64)
        k = int(r["num_verts"][j])
        assert k == len(keep) and np.array_equal(r["vert_ids"][j, :k], keep), j
        _, faces = bs.convex_hull(clouds[j][keep])
        m, com, _ = bs.hull_mass_properties_faces(clouds[j][keep], faces)
        assert abs(r["mass"][j] / m - 1.0) < 1e-10 and np.abs(r["com"][j] - com).max() < 1e-10


def test_degenerate_clouds():
    cube = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 2)], dtype=np.float64)
    extra = np.array([[0.5, 0.5, 0.0], [0.5, 0.0, 1.0], [0.5, 0.5, 1.0], [1.0, 1.0, 2.0]])  # face centres, an interior point, a duplicate corner
    r = bs.compile_clouds_device(np.concatenate([cube, extra]), [0, 12], DEV, density=1.0)
    assert abs(r["mass"][0] - 2.0) < 1e-12 and np.allclose(r["com"][0], [0.5, 0.5, 1.0]) and int(r["num_verts"][0]) == 8
    assert np.allclose(np.diag(r["inertia"][0]), [2.0 * (1 + 4) / 12, 2.0 * (1 + 4) / 12, 2.0 * (1 + 1) / 12])
    flat = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.0], [0.3, 0.3, 0]])
    with pytest.raises(ValueError, match="coplanar"):
        bs.compile_clouds_device(flat, [0, 5], DEV)
    with pytest.raises(ValueError, match="4 points"):
        bs.compile_clouds_device(flat, [0, 3, 5], DEV)
    with pytest.raises(RuntimeError, match="LDS"):
        bs.compile_clouds_device(np.random.default_rng(0).normal(size=(5000, 3)), [0, 5000], DEV)


def test_a_job_larger_than_max_points_is_refused_not_run():
    """advisor r5: the kernel sizes its LDS arrays from the caller's max_points; a job whose offsets disagree with it (more points, or a
    negative count) sets status 7 instead of overrunning them - the other jobs of the launch are compiled as usual."""
    from vid2player3d_amd import body_shapes as bs

    rng = np.random.default_rng(3)
    sizes = [40, 200, 40]
    pts = rng.normal(size=(sum(sizes), 3))
    off = np.concatenate([[0], np.cumsum(sizes)])
    with pytest.raises(ValueError, match="cloud 1 of 3.*max_points"):
        bs.compile_clouds_device(pts, off, DEV, max_points=64)
    out = bs.compile_clouds_device(pts, off, DEV)
    assert (out["num_verts"] >= 4).all()


def test_2048_shapes_compile_in_seconds():
    """The reference's scale: one body shape per AMASS clip.  2048 shapes = 49,152 hull jobs in one launch; the numpy loop takes ~0.4 s per
    shape (~15 min), the budget here is 30 s for clouds + device compile + BodyModel construction (VERDICT r4 #1a)."""
    base = load_baked_model()
    bs.synthetic_shape_family(base, 4, seed=0, device=DEV)  # (warm-up: library load, dense clouds of the base)
    t0 = time.perf_counter()
    clouds, rest = bs.deform_clouds(base, bs.family_params(2048, seed=7))
    t1 = time.perf_counter()
    fam = bs.bodies_from_clouds_device(base, clouds, rest, DEV)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("[shapes] 2048 shapes: clouds %.2f s, device compile + BodyModels %.2f s, total %.2f s" % (t1 - t0, t2 - t1, t2 - t0))
    assert len(fam) == 2048 and t2 - t0 < 30.0
    masses = np.array([m.total_mass for m in fam])
    assert masses.min() > 30 and masses.max() < 200 and np.all([np.diff(m.hull_offsets).max() <= 64 for m in fam[::97]])
    for s in (0, 1023, 2047):  # spot checks against the numpy statement
        _same_body(fam[s], bs.body_from_clouds(base, [c[s] for c in clouds], rest[s]), "shape %d of 2048" % s)


def test_smpl_betas_to_bodies_on_the_device():
    """bodies_from_smpl with a device: the clouds of a batch of betas through v2p_shapes_compile == the numpy path, shape by shape."""
    from tests.test_body_shapes import synthetic_smpl_model

    base = load_baked_model()
    smpl = synthetic_smpl_model(base)
    betas = np.random.default_rng(2).normal(0, 1.0, size=(6, 10))
    betas[0] = 0.0
    got = bs.bodies_from_smpl(smpl, betas, base, device=DEV)
    want = bs.bodies_from_smpl(smpl, betas, base)
    for s in range(6):
        _same_body(got[s], want[s], "betas %d" % s)
    assert abs(got[0].total_mass / base.total_mass - 1.0) < 1e-6
