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


def deform(base, leg=1.0, arm=1.0, trunk=1.0, girth=1.0, shoulder=1.0, belly=1.0, taper=0.0, bulge=0.0, **model_kw):
    """One non-uniform variant of `base`: bone offsets of the leg / arm / trunk chains scaled along their own direction, the
    clouds stretched by the same factor along the bone and by `girth` across it, shoulders moved apart by `shoulder`, the
    pelvis / torso / spine clouds inflated by `belly` - the kind of variation SMPL betas produce (height, limb proportions, weight).
    `taper` / `bulge` make the cross-section vary along the bone (linearly / quadratically): NON-affine, so the set of extreme
    points - the hull topology - changes, not only its coordinates."""
    clouds, rest = clouds_of(base, dense=(taper != 0.0 or bulge != 0.0))
    names = base.body_names
    nb = base.num_bodies
    fac = {n: (leg if n in _LEGS else arm if n in _ARMS else trunk) for n in names}
    new_local = base.local_pos.copy()
    for b in range(1, nb):
        new_local[b] = base.local_pos[b] * fac[names[b]]
        if names[b] in ("L_Thorax", "R_Thorax", "L_Shoulder", "R_Shoulder"):
            lat = np.zeros(3)
            k = int(np.argmax(np.abs(base.local_pos[base.body_index("L_Shoulder")])))  # the lateral axis of the rest pose
            lat[k] = 1.0
            new_local[b] = new_local[b] + (shoulder - 1.0) * (base.local_pos[b] @ lat) * lat
    new_rest = np.zeros((nb, 3))
    for b in range(nb):
        p = int(base.parents[b])
        new_rest[b] = new_local[b] + (new_rest[p] if p >= 0 else 0.0)
    children = base.children_lists()
    new_clouds = []
    for b in range(nb):
        # bone axis of body b: towards its first child (leaves: from the parent)
        axis = base.local_pos[children[b][0]] if children[b] else base.local_pos[b]
        axis = axis / (np.linalg.norm(axis) + 1e-12)
        along = clouds[b] @ axis
        across = clouds[b] - np.outer(along, axis)
        g = girth * (belly if names[b] in ("Pelvis", "Torso", "Spine") else 1.0)
        u = (along - along.min()) / (np.ptp(along) + 1e-12)  # 0 .. 1 along the bone
        prof = g * (1.0 + taper * (u - 0.5) + bulge * (0.25 - (u - 0.5) ** 2) * 4.0)
        new_clouds.append(np.outer(along * fac[names[b]], axis) + across * prof[:, None])
    return body_from_clouds(base, new_clouds, new_rest, **model_kw)


def synthetic_shape_family(base, num, seed=0, **model_kw):
    """`num` non-uniform variants of `base` (seeded): the stand-in for one SMPL shape per clip."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num):
        h = rng.uniform(0.9, 1.1)  # overall height
        out.append(deform(base, leg=h * rng.uniform(0.92, 1.08), arm=h * rng.uniform(0.92, 1.08), trunk=h * rng.uniform(0.95, 1.05),
                          girth=rng.uniform(0.82, 1.08), shoulder=rng.uniform(0.9, 1.15), belly=rng.uniform(0.9, 1.2),
                          taper=rng.uniform(-0.25, 0.25), bulge=rng.uniform(-0.15, 0.1), **model_kw))
    return out
