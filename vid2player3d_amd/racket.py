"""Racket + ball additions to the body model.

vid2player's `data/assets/smpl_mesh_humanoid_djokovic.xml:188-190` welds a body "Racket" (no joint) to R_Wrist at (-0.5, 0, 0) with two
cylinder geoms - the handle (radius 0.016, density 500, from (0.5,0,0) to (0.15,0,0) in the racket frame) and the head (radius 0.15,
density 150, a 4.2 cm thick disc whose axis is (0,1,1)/sqrt 2) -; `data/assets/tennis_ball.urdf` is a free sphere (radius 0.032,
mass 0.057, inertia 4e-5).  A body welded without a joint is part of its parent link for the dynamics: `with_racket` folds the
racket's mass, centre of mass and inertia into the wrist link of a BodyModel, appends a coarse vertex set of the two cylinders to
the wrist's hull vertices (racket-ground contact goes through the same hull-vertex rows as every other body) and returns the
cylinders in the wrist frame for the ball contacts.  The racket keeps its own row in the exported rigid-body state (index 24, like
Isaac Gym's rigid body tensor: `humanoid_smpl_im_mvae.py:68`).
"""
import numpy as np

from . import body_shapes
from .model import BodyModel

RACKET_PARENT = "R_Wrist"
RACKET_OFFSET = np.array([-0.5, 0.0, 0.0])  # body "Racket" pos in the wrist frame
# the three player assets (data/assets/smpl_mesh_humanoid_{djokovic,federer,nadal}.xml): the link the racket is welded to, the racket
# body's position in that link's frame, the handle's end points in the RACKET frame, and the joint ranges of the racket arm in degrees
# (djokovic :173, 178-180; federer :178 differs in Wrist_x; nadal :143, 148-150, 158-160 is left-handed: the mirror image along x)
PLAYERS = {
    "djokovic": {"parent": "R_Wrist", "offset": (-0.5, 0.0, 0.0), "handle": ((0.5, 0, 0), (0.15, 0, 0)),
                 "limits": {"R_Elbow": ((-180.0, 90.0), None, None), "R_Wrist": ((-10.0, 10.0), (-45.0, 45.0), (-90.0, 90.0))}},
    "federer": {"parent": "R_Wrist", "offset": (-0.5, 0.0, 0.0), "handle": ((0.5, 0, 0), (0.15, 0, 0)),
                "limits": {"R_Elbow": ((-180.0, 90.0), None, None), "R_Wrist": ((-90.0, 10.0), (-45.0, 45.0), (-90.0, 90.0))}},
    "nadal": {"parent": "L_Wrist", "offset": (0.5, 0.0, 0.0), "handle": ((-0.5, 0, 0), (-0.15, 0, 0)),
              "limits": {"L_Elbow": ((-180.0, 90.0), None, None), "L_Wrist": ((-90.0, 10.0), (-45.0, 45.0), (-90.0, 90.0))}},
}
BALL = {"radius": 0.032, "mass": 0.057, "inertia": 4e-5}  # tennis_ball.urdf
# contact material (humanoid_smpl_im_mvae.py:414-416, 436-438; plane: amass_im / djokovic yaml restitution 0, friction 1): PhysX combines
# the two shapes' values by averaging (its default combine mode)
BALL_MATERIAL = {"rest_ground": 0.5 * (1.0 + 0.0), "fric_ground": 0.5 * (0.8 + 1.0), "rest_racket": 0.5 * (1.0 + 1.0), "fric_racket": 0.5 * (0.8 + 0.8),
                 "rest_body": 0.5 * (1.0 + 0.0), "fric_body": 0.5 * (0.8 + 1.0),  # ball x a link's hull (shape defaults on the humanoid's side)
                 "bounce_threshold": 0.2, "ang_damp": 0.5, "max_ang_vel": 64.0}  # the last two: gymapi.AssetOptions defaults (the ball asset sets none)


def racket_cylinders(player="djokovic"):
    """[(centre, unit axis, half length, radius, density)] of handle and head in the WRIST frame."""
    out = []
    spec = PLAYERS[player]
    offset = np.array(spec["offset"], float)
    for a, b, radius, density in ((spec["handle"][0], spec["handle"][1], 0.016, 500.0), ((0, -0.015, -0.015), (0, 0.015, 0.015), 0.15, 150.0)):
        a, b = np.array(a, float) + offset, np.array(b, float) + offset
        axis = b - a
        out.append((0.5 * (a + b), axis / np.linalg.norm(axis), 0.5 * np.linalg.norm(axis), radius, density))
    return out


def _cylinder_mass_properties(centre, axis, half_len, radius, density):
    m = density * np.pi * radius ** 2 * 2 * half_len
    i_axis, i_perp = 0.5 * m * radius ** 2, m * (3 * radius ** 2 + (2 * half_len) ** 2) / 12.0
    aa = np.outer(axis, axis)
    return m, centre, i_axis * aa + i_perp * (np.eye(3) - aa)


def _cylinder_vertices(centre, axis, half_len, radius, n_ring):
    u = np.cross(axis, [1.0, 0.0, 0.0])
    if np.linalg.norm(u) < 1e-6:
        u = np.cross(axis, [0.0, 1.0, 0.0])
    u /= np.linalg.norm(u)
    w = np.cross(axis, u)
    ang = 2 * np.pi * np.arange(n_ring) / n_ring
    ring = radius * (np.outer(np.cos(ang), u) + np.outer(np.sin(ang), w))
    return np.concatenate([centre + half_len * axis + ring, centre - half_len * axis + ring])


# joint ranges of the racket arm in the djokovic MJCF, degrees (every other DOF is +-180 / +-720)
PLAYER_ARM_LIMITS = PLAYERS["djokovic"]["limits"]


def with_racket(base, wrist_vertex_budget=30, arm_limits=True, player="djokovic", **model_kw):
    """(BodyModel with the racket folded into the player's racket wrist, geometry dict for the ball contacts).  arm_limits: carry the
    player MJCF's joint ranges of the racket arm (enforced when cfg['env']['joint_limits'] is on).  player: djokovic / federer (right
    hand) or nadal (left hand: the reference's cfg_v2p righthand = False)."""
    spec = PLAYERS[player]
    if "racket_player" in base.blob:
        raise ValueError("this body model already carries a racket (%s): fold it into the plain body model, once" % str(base.blob["racket_player"]))
    b = base.body_index(spec["parent"])
    blob = dict(base.blob)
    blob["racket_player"] = np.asarray(player)
    if arm_limits:
        lo, hi = base.limit_lower.copy(), base.limit_upper.copy()
        for name, ranges in spec["limits"].items():
            j = 3 * (base.body_index(name) - 1)
            for i, rg in enumerate(ranges):
                if rg is not None:
                    lo[j + i], hi[j + i] = np.deg2rad(rg[0]), np.deg2rad(rg[1])
        blob["limit_lower"], blob["limit_upper"] = lo, hi
    cyls = racket_cylinders(player)
    # composite rigid body: wrist + handle + head
    parts = [(base.mass[b], base.com[b], base.inertia[b])] + [_cylinder_mass_properties(*c) for c in cyls]
    mass = sum(p[0] for p in parts)
    com = sum(p[0] * np.asarray(p[1]) for p in parts) / mass
    inertia = np.zeros((3, 3))
    for m, c, i in parts:
        d = np.asarray(c) - com
        inertia += np.asarray(i) + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    for k, v in (("mass", mass), ("com", com), ("inertia", inertia)):
        arr = np.array(blob[k], dtype=np.float64)
        arr[b] = v
        blob[k] = arr
    # contact vertices of the link: the wrist's own hull (thinned to make room) + the rims of the two cylinders
    off = np.asarray(blob["hull_offsets"])
    hv = np.asarray(blob["hull_verts"], dtype=np.float64)
    wrist = hv[off[b]:off[b + 1]]
    wrist = wrist[body_shapes.reduce_hull(wrist, wrist_vertex_budget)] if len(wrist) > wrist_vertex_budget else wrist
    rims = np.concatenate([_cylinder_vertices(cyls[0][0], cyls[0][1], cyls[0][2], cyls[0][3], 6), _cylinder_vertices(cyls[1][0], cyls[1][1], cyls[1][2], cyls[1][3], 10)])
    new = np.concatenate([wrist, rims])
    assert len(new) <= body_shapes.MAX_HULL_VERTS
    blob["hull_verts"] = np.concatenate([hv[:off[b]], new, hv[off[b + 1]:]])
    noff = off.copy()
    noff[b + 1:] += len(new) - (off[b + 1] - off[b])
    blob["hull_offsets"] = noff.astype(np.int32)
    geom = {"racket_link": b, "racket_offset": np.array(spec["offset"], float), "player": player,
            "cylinders": [{"center": c[0], "axis": c[1], "half_len": c[2], "radius": c[3]} for c in cyls],
            "racket_mass": float(sum(p[0] for p in parts[1:]))}
    # the gain / mass scales the base model was built with carry over (cfg env kp_scale, kd_scale, default_humanoid_mass)
    return BodyModel(blob, **{**getattr(base, "model_kw", {}), **model_kw}), geom
