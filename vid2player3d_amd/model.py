"""Body-model compiler: MJCF + binary-STL hulls -> flat structure-of-arrays model blob.

The reference never parses the body model itself: it hands the MJCF to Isaac Gym
(`gym.load_asset`, embodied_pose/env/tasks/humanoid_smpl_im.py:280-287) and PhysX cooks
convex hulls, integrates mass properties at the geom density and merges the three hinge
joints of every body into one spherical joint.  This module is the MI355X engine's
replacement for that importer.  It produces exactly what the HIP kernels need:

  parents[B]            kinematic tree (body order == rigid-body index)
  local_pos[B,3]        joint offset in the parent frame (MJCF `body pos`)
  mass[B] com[B,3] inertia[B,3,3]   hull-integrated at the geom density, inertia about COM
  kp[D] kd[D] armature[D]            per hinge axis (MJCF `stiffness/damping/armature`)
  hull_offsets[B+1] hull_verts[V,3]  convex-hull vertices per body (contact candidates)

`compile_mjcf` runs wherever the asset files exist; the result for the one body model the
reference ships (embodied_pose/data/assets/mjcf/smpl_mesh_humanoid_amass_v1.xml) is baked
into `vid2player3d_amd/data/` by `python -m vid2player3d_amd.model` so that the GPU box, which has no
reference checkout, loads the compiled blob.
"""
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
BAKED_MODEL = os.path.join(DATA_DIR, "smpl_humanoid_amass_v1.npz")

NUM_BODIES = 24
NUM_DOF = 69


def read_binary_stl(path):
    """Return the [T,3,3] triangle array of a binary STL file."""
    with open(path, "rb") as f:
        buf = f.read()
    (ntri,) = struct.unpack_from("<I", buf, 80)
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    tris = np.frombuffer(buf, dtype=rec, count=ntri, offset=84)
    return np.array(tris["v"], dtype=np.float64)


def hull_mass_properties(verts, density):
    """Mass, centre of mass and inertia-about-COM of the convex hull of `verts`.

    Signed-tetrahedron integration over the hull faces; faces are re-oriented outward
    here because the face winding stored in the STL files is not guaranteed.
    """
    from scipy.spatial import ConvexHull

    hull = ConvexHull(verts)
    centre = verts[hull.vertices].mean(axis=0)
    vol = 0.0
    first = np.zeros(3)
    second = np.zeros((3, 3))
    for simplex, eq in zip(hull.simplices, hull.equations):
        a, b, c = verts[simplex] - centre
        det = np.dot(a, np.cross(b, c))
        if np.dot(np.cross(b - a, c - a), eq[:3]) < 0:  # make the face wind outward
            det = -det
        vol += det / 6.0
        first += det / 24.0 * (a + b + c)
        s = a + b + c
        second += det / 120.0 * (np.outer(a, a) + np.outer(b, b) + np.outer(c, c) + np.outer(s, s))
    com_rel = first / vol
    mass = density * vol
    cov = density * second - mass * np.outer(com_rel, com_rel)  # covariance about COM
    inertia = np.trace(cov) * np.eye(3) - cov
    return mass, centre + com_rel, inertia, np.sort(hull.vertices)


def compile_mjcf(mjcf_path):
    """Parse an SMPL-humanoid MJCF (bodies with 3 hinge joints + one mesh geom each)."""
    root = ET.parse(mjcf_path).getroot()
    base = os.path.dirname(os.path.abspath(mjcf_path))
    mesh_files = {m.get("name"): os.path.normpath(os.path.join(base, m.get("file"))) for m in root.find("asset").findall("mesh")}
    default_joint = root.find("default").find("joint")
    def_arm = float(default_joint.get("armature", 0.0))

    names, parents, local_pos = [], [], []
    mass, com, inertia = [], [], []
    kp, kd, arm, lim_lo, lim_hi = [], [], [], [], []
    hull_verts, hull_offsets = [], [0]

    def visit(node, parent):
        idx = len(names)
        names.append(node.get("name"))
        parents.append(parent)
        local_pos.append([float(x) for x in node.get("pos").split()])
        q = [float(x) for x in node.get("quat", "1 0 0 0").split()]
        if not np.allclose(q, [1, 0, 0, 0]):
            raise ValueError("body %s: non-identity body quat is not supported" % names[-1])
        joints = node.findall("joint")
        if parent >= 0:
            if len(joints) != 3:
                raise ValueError("body %s: expected 3 hinge joints, got %d" % (names[-1], len(joints)))
            for j, ax in zip(joints, np.eye(3)):
                if j.get("type") != "hinge" or not np.allclose([float(x) for x in j.get("axis").split()], ax):
                    raise ValueError("body %s: joints must be x,y,z hinges" % names[-1])
                kp.append(float(j.get("stiffness", 0.0)))
                kd.append(float(j.get("damping", 0.0)))
                arm.append(float(j.get("armature", def_arm)))
                lo, hi = [float(x) for x in j.get("range", "-180 180").split()]
                lim_lo.append(np.deg2rad(lo))
                lim_hi.append(np.deg2rad(hi))
        geoms = node.findall("geom")
        if len(geoms) != 1 or geoms[0].get("type") != "mesh":
            raise ValueError("body %s: expected exactly one mesh geom" % names[-1])
        tris = read_binary_stl(mesh_files[geoms[0].get("mesh")])
        uniq = np.unique(tris.reshape(-1, 3), axis=0)
        m, c, inert, hv = hull_mass_properties(uniq, float(geoms[0].get("density", 1000.0)))
        mass.append(m)
        com.append(c)
        inertia.append(inert)
        hull_verts.append(uniq[hv])
        hull_offsets.append(hull_offsets[-1] + len(hv))
        for child in node.findall("body"):
            visit(child, idx)

    tops = root.find("worldbody").findall("body")
    if len(tops) != 1:
        raise ValueError("expected a single root body")
    visit(tops[0], -1)

    return {
        "body_names": np.array(names),
        "parents": np.array(parents, dtype=np.int32),
        "local_pos": np.array(local_pos, dtype=np.float64),
        "mass": np.array(mass, dtype=np.float64),
        "com": np.array(com, dtype=np.float64),
        "inertia": np.array(inertia, dtype=np.float64),
        "kp": np.array(kp, dtype=np.float64),
        "kd": np.array(kd, dtype=np.float64),
        "armature": np.array(arm, dtype=np.float64),
        "limit_lower": np.array(lim_lo, dtype=np.float64),
        "limit_upper": np.array(lim_hi, dtype=np.float64),
        "hull_offsets": np.array(hull_offsets, dtype=np.int32),
        "hull_verts": np.concatenate(hull_verts, axis=0).astype(np.float64),
    }


class BodyModel:
    """Flat body model + the derived quantities the task needs (gains scaled by body mass)."""

    def __init__(self, blob, default_humanoid_mass=90.0, kp_scale=1.0, kd_scale=1.0):
        self.blob = {k: np.asarray(v) for k, v in blob.items()}
        self.model_kw = dict(default_humanoid_mass=default_humanoid_mass, kp_scale=kp_scale, kd_scale=kd_scale)  # what derived models inherit
        self.body_names = [str(x) for x in self.blob["body_names"]]
        self.parents = self.blob["parents"].astype(np.int32)
        self.num_bodies = len(self.body_names)
        self.num_dof = 3 * (self.num_bodies - 1)
        self.local_pos = self.blob["local_pos"].astype(np.float64)
        self.mass = self.blob["mass"].astype(np.float64)
        self.com = self.blob["com"].astype(np.float64)
        self.inertia = self.blob["inertia"].astype(np.float64)
        self.total_mass = float(self.mass.sum())
        # gains are scaled by body mass / 90 exactly as the reference scales the Isaac Gym
        # dof properties (humanoid_smpl_im.py:376-385)
        pd_scale = self.total_mass / default_humanoid_mass
        self.kp = self.blob["kp"] * pd_scale * kp_scale
        self.kd = self.blob["kd"] * pd_scale * kd_scale
        self.armature = self.blob["armature"].astype(np.float64)
        # per-DOF range of the exponential-map coordinate, radians (MJCF `range`); +-pi = unlimited
        self.limit_lower = self.blob["limit_lower"].astype(np.float64) if "limit_lower" in self.blob else np.full(self.num_dof, -np.pi)
        self.limit_upper = self.blob["limit_upper"].astype(np.float64) if "limit_upper" in self.blob else np.full(self.num_dof, np.pi)
        self.hull_offsets = self.blob["hull_offsets"].astype(np.int32)
        self.hull_verts = self.blob["hull_verts"].astype(np.float64)
        # dof bookkeeping the reference derives from asset dof names (humanoid_smpl_im.py:159-178)
        self.dof_body_ids = list(range(1, self.num_bodies))
        self.dof_offsets = list(range(0, self.num_dof + 1, 3))

    def scaled(self, scale, **kw):
        """The same body uniformly scaled (the reference's per-clip `_motion_body_scales`, humanoid_smpl_im.py:255): lengths x s,
        masses x s^3, inertias x s^5; the gains follow the total mass like every asset's do (humanoid_smpl_im.py:376-385)."""
        s = float(scale)
        blob = dict(self.blob)
        blob["local_pos"] = self.blob["local_pos"] * s
        blob["com"] = self.blob["com"] * s
        blob["hull_verts"] = self.blob["hull_verts"] * s
        blob["mass"] = self.blob["mass"] * s ** 3
        blob["inertia"] = self.blob["inertia"] * s ** 5
        return BodyModel(blob, **{**self.model_kw, **kw})

    def body_index(self, name):
        return self.body_names.index(name)

    def children_lists(self):
        ch = [[] for _ in range(self.num_bodies)]
        for b, p in enumerate(self.parents):
            if p >= 0:
                ch[p].append(b)
        return ch


def load_baked_model(**kw):
    if not os.path.exists(BAKED_MODEL):
        raise FileNotFoundError("compiled body model missing: %s (run `python -m vid2player3d_amd.model`)" % BAKED_MODEL)
    with np.load(BAKED_MODEL, allow_pickle=False) as z:
        blob = {k: z[k] for k in z.files}
    return BodyModel(blob, **kw)


def main():
    import argparse

    ap = argparse.ArgumentParser(description="compile an SMPL-humanoid MJCF into the engine's model blob")
    ap.add_argument("--mjcf", default="/root/reference/embodied_pose/data/assets/mjcf/smpl_mesh_humanoid_amass_v1.xml")
    ap.add_argument("--out", default=BAKED_MODEL)
    args = ap.parse_args()
    blob = compile_mjcf(args.mjcf)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, **blob)
    m = BodyModel(blob)
    print("bodies %d  dof %d  hull verts %d  total mass %.3f kg" % (m.num_bodies, m.num_dof, len(m.hull_verts), m.total_mass))
    for n, ms in zip(m.body_names, m.mass):
        print("  %-12s %7.3f kg" % (n, ms))


if __name__ == "__main__":
    main()
