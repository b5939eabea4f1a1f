"""Per-clip body assets, geometry half (vid2player3d_amd/body_shapes.py): the own convex hull against scipy's qhull
(the reference's `ConvexHull`, smpl_local_robot.py:103), mass properties of hulls, hull reduction to the engine's vertex limit,
non-uniform shape variants."""
import numpy as np
import pytest
from scipy.spatial import ConvexHull

from vid2player3d_amd import body_shapes as bs
from vid2player3d_amd.model import hull_mass_properties, load_baked_model


@pytest.mark.parametrize("seed,n,aniso", [(0, 40, (1, 1, 1)), (1, 300, (1.0, 0.3, 2.0)), (2, 1500, (0.2, 0.2, 1.0)), (3, 8, (1, 1, 1))])
def test_convex_hull_matches_qhull(seed, n, aniso):
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, 3)) * np.array(aniso) + rng.normal(size=3)
    vid, faces = bs.convex_hull(pts)
    ref = ConvexHull(pts)
    assert np.array_equal(vid, np.sort(ref.vertices))
    assert len(faces) == len(ref.simplices)  # simplicial hulls of points in general position: same triangulation size
    # outward winding: every point is behind every face
    a, b, c = (pts[faces[:, k]] for k in range(3))
    nrm = np.cross(b - a, c - a)
    assert ((pts[None, :, :] - a[:, None, :]) * nrm[:, None, :]).sum(-1).max() < 1e-9
    m, com, inertia = bs.hull_mass_properties_faces(pts, faces, density=900.0)
    assert abs(m - 900.0 * ref.volume) < 1e-9 * m
    m2, com2, inertia2, _ = hull_mass_properties(pts, 900.0)  # the scipy-based integration the MJCF compiler uses
    assert np.allclose(com, com2, atol=1e-10) and np.allclose(inertia, inertia2, rtol=1e-9, atol=1e-12)


def test_hull_handles_coplanar_and_duplicate_points():
    cube = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 2)], dtype=np.float64)
    extra = np.array([[0.5, 0.5, 0.0], [0.5, 0.0, 1.0], [0.5, 0.5, 1.0], [1.0, 1.0, 2.0]])  # face centres, an interior point, a duplicate corner
    vid, faces = bs.convex_hull(np.concatenate([cube, extra]))
    assert set(vid.tolist()) <= set(range(8)) | {11} and len(set(map(tuple, np.concatenate([cube, extra])[vid]))) == 8
    m, com, inertia = bs.hull_mass_properties_faces(np.concatenate([cube, extra]), faces, density=1.0)
    assert abs(m - 2.0) < 1e-12 and np.allclose(com, [0.5, 0.5, 1.0])
    assert np.allclose(np.diag(inertia), [2.0 * (1 + 4) / 12, 2.0 * (1 + 4) / 12, 2.0 * (1 + 1) / 12])
    with pytest.raises(ValueError):
        bs.convex_hull(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.0], [0.3, 0.3, 0]]))


def test_reduce_hull_keeps_the_extent():
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(2000, 3)) * np.array([0.05, 0.08, 0.2])
    keep = bs.reduce_hull(pts, 64)
    assert len(keep) <= 64 and len(keep) > 40
    full = pts[bs.convex_hull(pts)[0]]
    d = bs.fibonacci_directions(500)
    assert ((full @ d.T).max(0) - (pts[keep] @ d.T).max(0)).max() < 0.012  # support function within ~1 cm on a 20 cm body
    v_full, v_red = ConvexHull(full).volume, ConvexHull(pts[keep]).volume
    assert 0.85 < v_red / v_full <= 1.0 + 1e-12


def test_identity_deformation_reproduces_the_baked_body():
    base = load_baked_model()
    same = bs.deform(base)
    assert np.allclose(same.local_pos, base.local_pos)
    assert abs(same.total_mass / base.total_mass - 1.0) < 1e-6
    assert np.allclose(same.com, base.com, atol=1e-6) and np.allclose(same.inertia, base.inertia, rtol=1e-5, atol=1e-9)
    assert np.array_equal(same.hull_offsets, base.hull_offsets)


def test_shape_family_is_non_uniform():
    base = load_baked_model()
    fam = bs.synthetic_shape_family(base, 6, seed=3)
    ratios, legs = [], []
    for m in fam:
        assert m.num_bodies == 24 and np.diff(m.hull_offsets).max() <= 64 and np.diff(m.hull_offsets).min() >= 4
        assert np.all(np.linalg.eigvalsh(m.inertia) > 0)
        ratios.append(m.mass[base.body_index("Torso")] / m.mass[base.body_index("L_Knee")])
        legs.append(np.linalg.norm(m.local_pos[base.body_index("L_Knee")]) / np.linalg.norm(m.local_pos[base.body_index("L_Elbow")]))
        assert abs(m.kp[0] / base.kp[0] - m.total_mass / base.total_mass) < 1e-9  # gains follow the mass (humanoid_smpl_im.py:376-385)
    # not copies of one body at different sizes: mass ratios between bodies and limb proportions change from shape to shape
    assert np.ptp(ratios) / np.mean(ratios) > 0.1 and np.ptp(legs) / np.mean(legs) > 0.03
    assert len({tuple(np.diff(m.hull_offsets)) for m in fam}) == len(fam)  # non-affine: every shape has its own hull topology
    uni = base.scaled(1.1)
    assert abs(uni.mass[9] / uni.mass[2] - base.mass[9] / base.mass[2]) < 1e-12  # (what uniform scaling cannot do)


def test_batched_family_equals_the_shape_by_shape_one():
    """family_params draws what the former per-shape loop drew (nine uniforms per shape, in order), and deform_clouds for S shapes at once
    gives each shape the clouds and rest joints `deform` builds for it alone."""
    base = load_baked_model()
    rng = np.random.default_rng(3)
    P = bs.family_params(4, seed=3)
    for i in range(4):
        h = rng.uniform(0.9, 1.1)
        row = dict(leg=h * rng.uniform(0.92, 1.08), arm=h * rng.uniform(0.92, 1.08), trunk=h * rng.uniform(0.95, 1.05), girth=rng.uniform(0.82, 1.08),
                   shoulder=rng.uniform(0.9, 1.15), belly=rng.uniform(0.9, 1.2), taper=rng.uniform(-0.25, 0.25), bulge=rng.uniform(-0.15, 0.1))
        assert all(P[k][i] == row[k] for k in row)
    clouds, rest = bs.deform_clouds(base, P)
    one, rest1 = bs.deform_clouds(base, {k: v[2:3] for k, v in P.items()})
    assert all(np.array_equal(c[2], o[0]) for c, o in zip(clouds, one)) and np.array_equal(rest[2], rest1[0])
    tabs, off = bs.reduce_direction_tables(64)
    assert list(np.diff(off)) == [256, 204, 163, 130, 104, 83, 66, 52] and tabs.shape == (off[-1], 3)


def synthetic_smpl_model(base, seed=0):
    """A model file of the SMPL FORMAT (v_template, shapedirs, J_regressor, weights) built from the baked body: its hull vertices placed at
    the rest joints are the skin, every vertex weighted 0.7 / 0.3 to its body's joint / the parent's; row j of the joint regressor is the
    minimum-norm affine combination of body j's vertices that gives its rest joint exactly; shape direction 0 = uniform growth by 5 % per
    unit, directions 1 .. 9 random."""
    clouds, rest = bs.clouds_of(base)
    names = list(base.body_names)
    vt, owner, span = [], [], {}
    for b, c in enumerate(clouds):
        j = bs.SMPL_JOINT_NAMES.index(names[b])
        span[j] = (len(owner), len(owner) + len(c))
        vt.append(c + rest[b])
        owner += [j] * len(c)
    vt, owner = np.concatenate(vt), np.array(owner)
    V = len(vt)
    w = np.zeros((V, 24))
    for j, n in enumerate(bs.SMPL_JOINT_NAMES):
        p = base.parents[names.index(n)]
        pj = bs.SMPL_JOINT_NAMES.index(names[p]) if p >= 0 else j
        a, e = span[j]
        w[a:e, j] += 0.7
        w[a:e, pj] += 0.3
    jr = np.zeros((24, V))
    for j, n in enumerate(bs.SMPL_JOINT_NAMES):
        a, e = span[j]
        A = np.concatenate([vt[a:e].T, np.ones((1, e - a))])                 # 4 equations: sum c v = joint, sum c = 1
        jr[j, a:e] = np.linalg.lstsq(A, np.concatenate([rest[names.index(n)], [1.0]]), rcond=None)[0]
    rng = np.random.default_rng(seed)
    sd = rng.normal(0, 0.004, size=(V, 3, 10))
    sd[:, :, 0] = 0.05 * vt
    return {"v_template": vt, "shapedirs": sd, "J_regressor": jr, "weights": w}


def test_smpl_betas_to_bodies(tmp_path):
    """SMPL(betas) -> vertex clouds -> bodies (smpl_parser.get_mesh_offsets + get_joint_geometries of the reference; the licensed model file
    is absent: a synthetic file of the same format).  betas = 0 gives back the body the file was built from; the first shape direction of
    this file is uniform growth, so betas = (2, 0, ...) must give the body uniformly scaled by 1.1; other directions deform it; the file
    round-trips through .npz and through a plain pickle (and a pickle that smuggles a callable in is refused)."""
    import pickle

    base = load_baked_model()
    smpl = synthetic_smpl_model(base)
    np.savez(tmp_path / "m.npz", **smpl)
    loaded = bs.load_smpl_model(str(tmp_path / "m.npz"))
    assert all(np.array_equal(loaded[k], smpl[k]) for k in smpl)
    with open(tmp_path / "m.pkl", "wb") as f:
        pickle.dump({k: v for k, v in smpl.items()}, f, protocol=2)
    assert np.array_equal(bs.load_smpl_model(str(tmp_path / "m.pkl"))["weights"], smpl["weights"])
    with open(tmp_path / "evil.pkl", "wb") as f:
        pickle.dump({"v_template": print}, f)
    with pytest.raises(pickle.UnpicklingError):
        bs.load_smpl_model(str(tmp_path / "evil.pkl"))

    betas = np.zeros((3, 10))
    betas[1, 0] = 2.0
    betas[2, 1:] = np.random.default_rng(1).normal(0, 1.0, size=9)
    same, grown, other = bs.bodies_from_smpl(smpl, betas, base)
    assert np.allclose(same.local_pos, base.local_pos, atol=1e-10) and np.array_equal(same.hull_offsets, base.hull_offsets)
    assert abs(same.total_mass / base.total_mass - 1.0) < 1e-6 and np.allclose(same.com, base.com, atol=1e-6)
    want = base.scaled(1.1)
    assert np.allclose(grown.local_pos, want.local_pos, atol=1e-10) and np.allclose(grown.mass, want.mass, rtol=1e-6)
    assert np.allclose(grown.inertia, want.inertia, rtol=1e-5, atol=1e-10) and np.allclose(grown.kp, want.kp, rtol=1e-6)
    assert abs(other.total_mass / base.total_mass - 1.0) > 1e-3 and not np.allclose(other.local_pos, base.local_pos, atol=1e-4)
    assert np.all(np.linalg.eigvalsh(other.inertia) > 0) and np.diff(other.hull_offsets).max() <= 64
    # the options of get_mesh_offsets: a global scale, flat feet (the lowest centimetre of the skin levelled)
    c0, r0 = bs.smpl_clouds(smpl, betas[:1], base)
    c1, r1 = bs.smpl_clouds(smpl, betas[:1], base, scale=1.2)
    assert np.allclose(r1, 1.2 * r0) and all(np.allclose(a, 1.2 * b) for a, b in zip(c1, c0))
    c2, _ = bs.smpl_clouds(smpl, betas[:1], base, flatfoot=True)
    toe = base.body_index("L_Toe")
    low0, low2 = np.sort(c0[toe][0][:, 1])[:3], np.sort(c2[toe][0][:, 1])[:3]
    assert low2.min() >= low0.min() - 1e-12 and np.ptp(low2[:2]) <= np.ptp(low0[:2]) + 1e-12
