"""Per-clip body assets, geometry half.

The reference builds one humanoid asset per sampled clip from the clip's SMPL shape (`humanoid_smpl_im.py:255-296` ->
`uhc/smpllib/smpl_local_robot.py:1172-1456`): SMPL vertices are assigned to the joint with the largest skinning weight, each
body's vertex cloud (relative to its joint) becomes a convex hull (`get_joint_geometries`, `smpl_local_robot.py:79-143`:
scipy ConvexHull -> STL -> quadric decimation down to >= 50 vertices), Isaac Gym integrates mass properties at the geom
density 900 and scales the drive gains with the total mass (`humanoid_smpl_im.py:376-385`).

The licensed SMPL model (vertices / skinning weights from betas) is not redistributable and absent here; everything after it is
this module:

    clouds[b] ([n_b, 3], body-b joint frame) + rest joints [24, 3]
        -> convex hull of every cloud           (`convex_hull`: incremental, own implementation; tested against scipy's qhull)
        -> reduced to <= 64 support vertices     (the engine's per-body limit; the reference decimates to >= 50)
        -> mass, centre of mass, inertia at the geom density (signed tetrahedra over the hull faces)
        -> BodyModel (= one DevShape of the engine: `v2p_env_create_shapes`)

`synthetic_shape_family` stands in for "SMPL(betas)": it deforms the baked body's clouds and skeleton NON-uniformly (limb
lengths, girths, torso / leg proportions, shoulder width per shape), so that hull topology, mass ratios and inertia tensors
really differ between shapes - which uniformly scaled copies (`BodyModel.scaled`) never exercise.
"""
import numpy as np

from .model import BodyModel

GEOM_DENSITY = 900.0  # smpl_mesh_humanoid_amass_v1.xml:50-... (every geom)
MAX_HULL_VERTS = 64   # per body, the engine's limit (csrc/capi.hip v2p_model_create)


def convex_hull(points, eps_rel=1e-10):
    """3-D convex hull by incremental insertion.  Returns (vertex_ids sorted, faces [F,3] of point indices, wound outward).
    Points within eps of a face plane are treated as inside (coplanar interior points are not hull vertices)."""
    p = np.asarray(points, dtype=np.float64)
    n = len(p)
    if n < 4:
        raise ValueError("convex_hull needs at least 4 points")
    scale = np.abs(p - p.mean(0)).max() + 1e-300
    eps = eps_rel * scale
    # initial tetrahedron: extreme pair, farthest from their line, farthest from their plane
    i0 = int(np.argmin(p[:, 0]))
    i1 = int(np.argmax(np.linalg.norm(p - p[i0], axis=1)))
    d = p[i1] - p[i0]
    i2 = int(np.argmax(np.linalg.norm(np.cross(p - p[i0], d), axis=1)))
    nrm = np.cross(p[i1] - p[i0], p[i2] - p[i0])
    dist = (p - p[i0]) @ nrm
    i3 = int(np.argmax(np.abs(dist)))
    if abs(dist[i3]) <= eps * np.linalg.norm(nrm):
        raise ValueError("convex_hull: the points are coplanar")
    if dist[i3] > 0:
        i1, i2 = i2, i1  # make (i0, i1, i2) face away from i3
    faces = np.array([(i0, i1, i2), (i0, i3, i1), (i1, i3, i2), (i2, i3, i0)], dtype=np.int64)

    def planes_of(f):
        a, b, c = p[f[:, 0]], p[f[:, 1]], p[f[:, 2]]
        u, v = b - a, c - a
        nn = np.stack([u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1], u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2], u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]], axis=1)
        nn /= np.sqrt((nn * nn).sum(1, keepdims=True))
        return nn, (nn * a).sum(1)

    N, D = planes_of(faces)
    remaining = np.array([i for i in range(n) if i not in (i0, i1, i2, i3)], dtype=np.int64)
    while len(remaining):
        # signed distance of every remaining point to every face: points outside no face are inside the hull for good
        dist = p[remaining] @ N.T - D
        far = dist.max(1)
        keep = far > eps
        remaining, dist, far = remaining[keep], dist[keep], far[keep]
        if not len(remaining):
            break
        j = int(np.argmax(far))  # insert the point farthest outside (keeps the face count small)
        i, vis = int(remaining[j]), dist[j] > eps
        remaining = np.delete(remaining, j)
        # horizon: directed edges of visible faces whose reverse edge belongs to no visible face
        vf = faces[vis]
        e = np.concatenate([vf[:, [0, 1]], vf[:, [1, 2]], vf[:, [2, 0]]])
        key = e[:, 0] * n + e[:, 1]
        horizon = e[~np.isin(e[:, 1] * n + e[:, 0], key)]
        new = np.concatenate([horizon, np.full((len(horizon), 1), i, dtype=np.int64)], axis=1)
        nN, nD = planes_of(new)
        faces = np.concatenate([faces[~vis], new])
        N, D = np.concatenate([N[~vis], nN]), np.concatenate([D[~vis], nD])
    return np.unique(faces), faces


def hull_mass_properties_faces(points, faces, density=GEOM_DENSITY):
    """Mass, centre of mass, inertia about the COM (body axes) of the solid bounded by outward-wound triangles."""
    p = np.asarray(points, dtype=np.float64)
    centre = p[np.unique(faces)].mean(0)
    a, b, c = (p[faces[:, k]] - centre for k in range(3))
    det = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = det.sum() / 6.0
    first = (det[:, None] / 24.0 * (a + b + c)).sum(0)
    s = a + b + c
    second = np.einsum("f,fij->ij", det / 120.0, a[:, :, None] * a[:, None, :] + b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :] + s[:, :, None] * s[:, None, :])
    com_rel = first / vol
    mass = density * vol
    cov = density * second - mass * np.outer(com_rel, com_rel)
    return mass, centre + com_rel, np.trace(cov) * np.eye(3) - cov


def fibonacci_directions(k):
    i = np.arange(k) + 0.5
    z = 1.0 - 2.0 * i / k
    r = np.sqrt(np.clip(1.0 - z * z, 0.0, None))
    phi = i * np.pi * (3.0 - np.sqrt(5.0))
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


def reduce_hull(points, max_verts=MAX_HULL_VERTS):
    """At most `max_verts` hull vertices that keep the extent of the cloud in every direction: the support points of a sphere of
    directions (the hull of support points is inscribed in the full hull and touches it in those directions; the reference reaches
    a similar vertex count by quadric decimation of the hull mesh, smpl_local_robot.py:133-139).  Returns point indices (sorted)."""
    p = np.asarray(points, dtype=np.float64)
    vid, _ = convex_hull(p)
    if len(vid) <= max_verts:
        return vid
    c = p[vid] - p[vid].mean(0)
    k = 4 * max_verts
    while True:
        sel = np.unique(np.argmax(c @ fibonacci_directions(k).T, axis=0))
        if len(sel) <= max_verts:
            return vid[sel]
        k = int(k * 0.8)


def body_from_clouds(base, clouds, rest_joints, density=GEOM_DENSITY, max_verts=MAX_HULL_VERTS, **model_kw):
    """BodyModel (one engine DevShape) from per-body vertex clouds (body-joint frames, like `smpl_verts[vind] - smpl_jts[jind]`) and the
    rest joint positions [24,3] of the same shape; tree, joint gains / armature are `base`'s (the reference's MJCF template:
    skeleton_mesh_local.py:9-33), the gains then follow the new total mass exactly as every asset's do."""
    nb = base.num_bodies
    if len(clouds) != nb or np.shape(rest_joints) != (nb, 3):
        raise ValueError("expected %d clouds and rest joints [%d,3]" % (nb, nb))
    rest = np.asarray(rest_joints, dtype=np.float64)
    blob = dict(base.blob)
    local_pos = np.zeros((nb, 3))
    mass, com, inertia, hv, off = np.zeros(nb), np.zeros((nb, 3)), np.zeros((nb, 3, 3)), [], [0]
    for b in range(nb):
        par = int(base.parents[b])
        local_pos[b] = rest[b] - (rest[par] if par >= 0 else 0.0)
        pts = np.asarray(clouds[b], dtype=np.float64)
        keep = reduce_hull(pts, max_verts)
        sub = pts[keep]
        _, faces = convex_hull(sub)  # the simulated solid is the hull of the kept vertices (what contact sees is what has mass)
        mass[b], com[b], inertia[b] = hull_mass_properties_faces(sub, faces, density)
        hv.append(sub)
        off.append(off[-1] + len(sub))
    blob.update(local_pos=local_pos, mass=mass, com=com, inertia=inertia, hull_offsets=np.array(off, dtype=np.int32), hull_verts=np.concatenate(hv, 0))
    return BodyModel(blob, **model_kw)


def clouds_of(model, dense=False):
    """The baked body's hull vertices per body (body-joint frames) and its rest joints: the 'mean shape' cloud set.  dense: plus
    the edge midpoints and face centroids of every hull (a surface sampling, ~4x the points: under a non-affine deformation
    some of them become extreme points, as the skin vertices of a real body do)."""
    rest = np.zeros((model.num_bodies, 3))
    for b in range(model.num_bodies):
        p = int(model.parents[b])
        rest[b] = model.local_pos[b] + (rest[p] if p >= 0 else 0.0)
    clouds = [model.hull_verts[model.hull_offsets[b]:model.hull_offsets[b + 1]].copy() for b in range(model.num_bodies)]
    if dense:
        for b, v in enumerate(clouds):
            _, f = convex_hull(v)
            mids = np.concatenate([0.5 * (v[f[:, i]] + v[f[:, (i + 1) % 3]]) for i in range(3)])
            clouds[b] = np.concatenate([v, np.unique(np.round(mids, 12), axis=0), v[f].mean(1)])
    return clouds, rest


# limb groups of the SMPL tree (body order of SURVEY a14)
_LEGS = ("L_Hip", "L_Knee", "L_Ankle", "L_Toe", "R_Hip", "R_Knee", "R_Ankle", "R_Toe")
_ARMS = ("L_Thorax", "L_Shoulder", "L_Elbow", "L_Wrist", "L_Hand", "R_Thorax", "R_Shoulder", "R_Elbow", "R_Wrist", "R_Hand")
_TRUNK = ("Pelvis", "Torso", "Spine", "Chest", "Neck", "Head")


FAMILY_PARAMS = ("leg", "arm", "trunk", "girth", "shoulder", "belly", "taper", "bulge")
_DENSE_CACHE = {}


def _dense_clouds(base):
    """clouds_of(base, dense=True), once per base model (shape independent: every variant deforms the same sampled surface)."""
    key = id(base)
    if key not in _DENSE_CACHE or _DENSE_CACHE[key][0] is not base:
        _DENSE_CACHE.clear()
        _DENSE_CACHE[key] = (base, clouds_of(base, dense=True))
    return _DENSE_CACHE[key][1]


def deform_clouds(base, params, dense=True):
    """The vertex clouds and rest joints of S non-uniform variants of `base` at once: params = dict of [S] arrays (FAMILY_PARAMS).
    Bone offsets of the leg / arm / trunk chains are scaled along their own direction, the clouds stretched by the same factor along
    the bone and by `girth` across it, shoulders moved apart by `shoulder`, the pelvis / torso / spine clouds inflated by `belly` -
    the kind of variation SMPL betas produce (height, limb proportions, weight).  `taper` / `bulge` make the cross-section vary along
    the bone (linearly / quadratically): NON-affine, so the set of extreme points - the hull topology - changes, not only its
    coordinates.  Returns (clouds: list over bodies of [S, n_b, 3], rest joints [S, 24, 3])."""
    P = {k: np.atleast_1d(np.asarray(params.get(k, 0.0 if k in ("taper", "bulge") else 1.0), dtype=np.float64)) for k in FAMILY_PARAMS}
    S = max(len(v) for v in P.values())
    P = {k: np.broadcast_to(v, (S,)) for k, v in P.items()}
    clouds, _ = _dense_clouds(base) if dense else clouds_of(base)
    names = base.body_names
    nb = base.num_bodies
    fac = np.stack([P["leg"] if n in _LEGS else P["arm"] if n in _ARMS else P["trunk"] for n in names], axis=1)  # [S, nb]
    new_local = base.local_pos[None] * fac[:, :, None]
    lat = np.zeros(3)
    lat[int(np.argmax(np.abs(base.local_pos[base.body_index("L_Shoulder")])))] = 1.0  # the lateral axis of the rest pose
    for b in range(1, nb):
        if names[b] in ("L_Thorax", "R_Thorax", "L_Shoulder", "R_Shoulder"):
            new_local[:, b] = new_local[:, b] + ((P["shoulder"] - 1.0) * (base.local_pos[b] @ lat))[:, None] * lat
    new_local[:, 0] = base.local_pos[0]
    new_rest = np.zeros((S, nb, 3))
    for b in range(nb):
        p = int(base.parents[b])
        new_rest[:, b] = new_local[:, b] + (new_rest[:, p] if p >= 0 else 0.0)
    children = base.children_lists()
    out = []
    for b in range(nb):
        # bone axis of body b: towards its first child (leaves: from the parent)
        axis = base.local_pos[children[b][0]] if children[b] else base.local_pos[b]
        axis = axis / (np.linalg.norm(axis) + 1e-12)
        along = clouds[b] @ axis
        across = clouds[b] - np.outer(along, axis)
        g = P["girth"] * (P["belly"] if names[b] in ("Pelvis", "Torso", "Spine") else 1.0)
        u = (along - along.min()) / (np.ptp(along) + 1e-12)  # 0 .. 1 along the bone
        prof = g[:, None] * (1.0 + P["taper"][:, None] * (u - 0.5)[None] + P["bulge"][:, None] * ((0.25 - (u - 0.5) ** 2) * 4.0)[None])
        out.append((along[None, :] * fac[:, b, None])[:, :, None] * axis + across[None] * prof[:, :, None])
    return out, new_rest


def deform(base, leg=1.0, arm=1.0, trunk=1.0, girth=1.0, shoulder=1.0, belly=1.0, taper=0.0, bulge=0.0, **model_kw):
    """One non-uniform variant of `base` (deform_clouds for a single shape, compiled on the CPU)."""
    clouds, rest = deform_clouds(base, dict(leg=leg, arm=arm, trunk=trunk, girth=girth, shoulder=shoulder, belly=belly, taper=taper, bulge=bulge),
                                 dense=(taper != 0.0 or bulge != 0.0))
    return body_from_clouds(base, [c[0] for c in clouds], rest[0], **model_kw)


def family_params(num, seed=0):
    """The seeded parameters of `num` synthetic body shapes (one row of nine uniform draws per shape)."""
    u = np.random.default_rng(seed).random((num, 9))
    lohi = [(0.9, 1.1), (0.92, 1.08), (0.92, 1.08), (0.95, 1.05), (0.82, 1.08), (0.9, 1.15), (0.9, 1.2), (-0.25, 0.25), (-0.15, 0.1)]
    v = [lo + (hi - lo) * u[:, k] for k, (lo, hi) in enumerate(lohi)]
    h = v[0]  # overall height
    return dict(leg=h * v[1], arm=h * v[2], trunk=h * v[3], girth=v[4], shoulder=v[5], belly=v[6], taper=v[7], bulge=v[8])


def synthetic_shape_family(base, num, seed=0, device=None, **model_kw):
    """`num` non-uniform variants of `base` (seeded): the stand-in for one SMPL shape per clip.  device = a torch cuda device: hulls and
    mass properties of all num x 24 bodies in one launch of the engine's shape compiler (v2p_shapes_compile) instead of the numpy loop
    (0.4 s per shape) - same algorithm, same result to float64 rounding (tests/test_gpu_shapes.py)."""
    clouds, rest = deform_clouds(base, family_params(num, seed))
    if device is not None:
        return bodies_from_clouds_device(base, clouds, rest, device, **model_kw)
    return [body_from_clouds(base, [c[s] for c in clouds], rest[s], **model_kw) for s in range(num)]


def reduce_direction_tables(max_verts=MAX_HULL_VERTS):
    """The direction sets reduce_hull tries in turn (4 x max_verts Fibonacci directions, then 0.8 x as many, ... until a set no larger
    than max_verts, whose distinct support points always fit): (dirs [sum k, 3], offsets [tables + 1])."""
    ks, k = [], 4 * max_verts
    while True:
        ks.append(k)
        if k <= max_verts:
            break
        k = int(k * 0.8)
    return np.concatenate([fibonacci_directions(k) for k in ks]), np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)


def compile_clouds_device(points, offsets, device, density=GEOM_DENSITY, max_verts=MAX_HULL_VERTS, eps_rel=1e-10, max_points=None):
    """The engine's shape compiler on a flat list of clouds: points [T,3] float64, offsets [J+1] -> dict of numpy arrays per job
    (mass [J], com [J,3], inertia [J,3,3], num_verts [J], vert_ids [J,max_verts], verts [J,max_verts,3]).  HIP only (no CPU path:
    body_from_clouds is the numpy statement of the same algorithm and the checker of this one)."""
    import torch

    from . import _lib

    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("compile_clouds_device runs on the HIP engine only (device=%r)" % (device,))
    offsets = np.asarray(offsets, dtype=np.int64)
    J = len(offsets) - 1
    sizes = np.diff(offsets)
    if J < 1 or sizes.min() < 4:
        raise ValueError("every cloud needs at least 4 points")
    dirs, doff = reduce_direction_tables(max_verts)
    with torch.cuda.device(dev):
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)  # noqa: E731
        pts, off = t(points, torch.float64), t(offsets, torch.int32)
        d_dirs, d_doff = t(dirs, torch.float64), t(doff, torch.int32)
        mass = torch.zeros(J, dtype=torch.float64, device=dev)
        com = torch.zeros((J, 3), dtype=torch.float64, device=dev)
        inertia = torch.zeros((J, 3, 3), dtype=torch.float64, device=dev)
        nv = torch.zeros(J, dtype=torch.int32, device=dev)
        vid = torch.zeros((J, max_verts), dtype=torch.int32, device=dev)
        verts = torch.zeros((J, max_verts, 3), dtype=torch.float64, device=dev)
        status = torch.full((J,), -1, dtype=torch.int32, device=dev)
        _lib.check(lib.v2p_shapes_compile(J, _lib.ptr(pts), _lib.ptr(off), int(sizes.max() if max_points is None else max_points), _lib.ptr(d_dirs), _lib.ptr(d_doff), len(doff) - 1, float(density),
                                          int(max_verts), float(eps_rel), _lib.ptr(mass), _lib.ptr(com), _lib.ptr(inertia), _lib.ptr(nv), _lib.ptr(vid),
                                          _lib.ptr(verts), _lib.ptr(status), _lib.current_stream(dev)), "v2p_shapes_compile")
        st = status.cpu().numpy()
    if (st != 0).any():
        j = int(np.nonzero(st)[0][0])
        raise ValueError("v2p_shapes_compile: cloud %d of %d: %s" % (j, J, {1: "fewer than 4 points", 2: "the points are coplanar", 3: "hull face capacity exceeded",
                                                                          4: "support reduction failed", 7: "the job's point count is negative or exceeds max_points"}.get(int(st[j]), "status %d" % st[j])))
    return dict(mass=mass.cpu().numpy(), com=com.cpu().numpy(), inertia=inertia.cpu().numpy(), num_verts=nv.cpu().numpy(), vert_ids=vid.cpu().numpy(),
                verts=verts.cpu().numpy())


def bodies_from_clouds_device(base, clouds, rest_joints, device, density=GEOM_DENSITY, max_verts=MAX_HULL_VERTS, **model_kw):
    """body_from_clouds for S shapes at once on the device: clouds = list over the 24 bodies of [S, n_b, 3] (or of lists of [n, 3], one
    per shape), rest_joints [S, 24, 3] -> S BodyModels."""
    nb = base.num_bodies
    rest = np.asarray(rest_joints, dtype=np.float64)
    S = rest.shape[0]
    if len(clouds) != nb or rest.shape != (S, nb, 3):
        raise ValueError("expected %d cloud sets and rest joints [S,%d,3]" % (nb, nb))
    per = [[np.asarray(clouds[b][s], dtype=np.float64) for b in range(nb)] for s in range(S)]  # job order = (shape, body)
    sizes = np.array([[len(c) for c in row] for row in per]).reshape(-1)
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    points = np.concatenate([c for row in per for c in row], axis=0)
    r = compile_clouds_device(points, offsets, device, density, max_verts)
    par = np.asarray(base.parents)
    local_pos = rest - np.where(par[None, :, None] >= 0, rest[:, np.maximum(par, 0)], 0.0)
    out = []
    nv = r["num_verts"].reshape(S, nb)
    for s in range(S):
        j0 = s * nb
        blob = dict(base.blob)
        off = np.concatenate([[0], np.cumsum(nv[s])]).astype(np.int32)
        hv = np.concatenate([r["verts"][j0 + b, :nv[s, b]] for b in range(nb)], axis=0)
        blob.update(local_pos=local_pos[s], mass=r["mass"][j0:j0 + nb], com=r["com"][j0:j0 + nb], inertia=r["inertia"][j0:j0 + nb], hull_offsets=off, hull_verts=hv)
        out.append(BodyModel(blob, **model_kw))
    return out


# ---- SMPL(betas) -> per-body vertex clouds -----------------------------------------------------------------------------------------------
# What the reference does per clip before the hulls (uhc/smpllib/smpl_local_robot.py:1232-1251 -> smpl_parser.py:413-458 `get_mesh_offsets`,
# then `get_joint_geometries`, smpl_local_robot.py:79-101): the SMPL body of the clip's betas in the ZERO pose - vertices = v_template +
# shapedirs . betas (pose blend shapes and skinning vanish at zero pose), joints = J_regressor . vertices - and every vertex handed to the
# joint with the largest skinning weight, relative to that joint.  The SMPL model FILE is licensed and not shipped; this is the code that
# consumes it (any container of its four standard arrays), tested on a synthetic model of the same format (tests/test_body_shapes.py).
SMPL_JOINT_NAMES = ("Pelvis", "L_Hip", "R_Hip", "Torso", "L_Knee", "R_Knee", "Spine", "L_Ankle", "R_Ankle", "Chest", "L_Toe", "R_Toe", "Neck",
                    "L_Thorax", "R_Thorax", "Head", "L_Shoulder", "R_Shoulder", "L_Elbow", "R_Elbow", "L_Wrist", "R_Wrist", "L_Hand", "R_Hand")  # smpl_parser.py:10-35


def load_smpl_model(path):
    """{v_template [V,3], shapedirs [V,3,K], J_regressor [24+,V], weights [V,24]} from an .npz, or from a pickle of plain numpy /
    scipy-sparse arrays (the SMPL release's .pkl with its chumpy objects converted: `np.array(x)`); read through the restricted unpickler."""
    if str(path).endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            d = {k: z[k] for k in z.files}
    else:
        import pickle

        allowed = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
                   ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("scipy.sparse.csc", "csc_matrix"),
                   ("scipy.sparse._csc", "csc_matrix"), ("scipy.sparse.csr", "csr_matrix"), ("scipy.sparse._csr", "csr_matrix"),
                   ("_codecs", "encode")}  # (protocol-2 numpy pickles carry their bytes as a latin-1 string + _codecs.encode)

        class _Plain(pickle.Unpickler):  # arrays and sparse matrices only: a model file from the internet cannot run code
            def find_class(self, module, name):
                if (module, name) in allowed:
                    return super().find_class(module, name)
                raise pickle.UnpicklingError("SMPL model file: refusing to unpickle %s.%s (convert chumpy objects with np.array first)" % (module, name))

        with open(path, "rb") as f:
            d = _Plain(f, encoding="latin1").load()
    out = {}
    for k in ("v_template", "shapedirs", "J_regressor", "weights"):
        if k not in d:
            raise KeyError("%s: not an SMPL model (no %r)" % (path, k))
        v = d[k]
        out[k] = np.asarray(v.toarray() if hasattr(v, "toarray") else v, dtype=np.float64)
    return out


def smpl_clouds(smpl, betas, base, scale=None, flatfoot=False, joint_names=SMPL_JOINT_NAMES):
    """Per-body vertex clouds (body-joint frames) and rest joints of S SMPL shapes, in `base`'s MJCF body order: betas [S,K] ->
    (list over the 24 bodies of [S, n_b, 3], rest joints [S, 24, 3]).  scale [S] / flatfoot: the options of get_mesh_offsets
    (smpl_parser.py:424-429)."""
    betas = np.atleast_2d(np.asarray(betas, dtype=np.float64))
    S, K = betas.shape
    vt, sd, jr, w = smpl["v_template"], smpl["shapedirs"], smpl["J_regressor"], smpl["weights"]
    if sd.shape[2] < K or w.shape[1] < len(joint_names) or jr.shape[0] < len(joint_names):
        raise ValueError("SMPL model with %d shape directions / %d joints; need %d / %d" % (sd.shape[2], w.shape[1], K, len(joint_names)))
    verts = vt[None] + np.einsum("vck,sk->svc", sd[:, :, :K], betas)          # [S,V,3]
    joints = np.einsum("jv,svc->sjc", jr[:len(joint_names)], verts)           # (from the un-flattened vertices, like the reference)
    if scale is not None:
        sc = np.broadcast_to(np.asarray(scale, dtype=np.float64), (S,))
        verts, joints = verts * sc[:, None, None], joints * sc[:, None, None]
    if flatfoot:
        for s in range(S):
            feet = verts[s, :, 1] < verts[s, :, 1].min() + 0.01
            verts[s, feet, 1] = verts[s, feet, 1].mean()
    owner = w[:, :len(joint_names)].argmax(axis=1)                            # vertex -> SMPL joint (skin_weights.argmax, smpl_local_robot.py:91)
    names = list(base.body_names)
    clouds, rest = [], np.zeros((S, len(names), 3))
    for b, name in enumerate(names):
        j = joint_names.index(name)
        vind = np.nonzero(owner == j)[0]
        if len(vind) < 4:
            raise ValueError("SMPL joint %s owns %d vertices: no hull" % (name, len(vind)))
        clouds.append(verts[:, vind] - joints[:, j:j + 1])
        rest[:, b] = joints[:, j]
    return clouds, rest


def bodies_from_smpl(smpl, betas, base, device=None, scale=None, flatfoot=False, **model_kw):
    """One BodyModel per row of `betas`: the per-clip humanoid assets of the reference (humanoid_smpl_im.py:255-296), without the MJCF / STL
    files in between.  device = a torch cuda device: all hulls in one launch of v2p_shapes_compile."""
    clouds, rest = smpl_clouds(smpl, betas, base, scale, flatfoot)
    if device is not None:
        return bodies_from_clouds_device(base, clouds, rest, device, **model_kw)
    return [body_from_clouds(base, [c[s] for c in clouds], rest[s], **model_kw) for s in range(len(rest))]
