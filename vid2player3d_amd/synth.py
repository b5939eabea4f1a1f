"""Seeded synthetic reference-motion clips (AMASS is not redistributable).

A clip is what `uhc/utils/convert_amass_isaac.py:128-140` feeds to poselib: per-frame local
joint rotations (xyzw quaternions, MJCF body order, root carrying the SMPL y-up -> z-up base
rotation [.5,.5,.5,.5]) plus the root translation, at 30 fps.  The motion is a smooth random
walk: low-pass filtered joint axis-angles, a wandering heading and a gently swaying pelvis.
Pure numpy so that the same clips can be rebuilt on the GPU box for `bench.py`.
"""
import numpy as np

BASE_ROT = np.array([0.5, 0.5, 0.5, 0.5])  # xyzw, SMPL y-up body in a z-up world


def _quat_mul(a, b):
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
        w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2,
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
    ], axis=-1)


def _quat_from_rotvec(v):
    ang = np.linalg.norm(v, axis=-1, keepdims=True)
    half = 0.5 * ang
    k = np.where(ang > 1e-8, np.sin(half) / np.maximum(ang, 1e-8), 0.5)
    return np.concatenate([v * k, np.cos(half)], axis=-1)


def _smooth_walk(rng, n, dim, sigma, smooth):
    """Random walk low-passed with a box filter of width `smooth` frames."""
    steps = rng.normal(0.0, sigma, size=(n + 2 * smooth, dim))
    walk = np.cumsum(steps, axis=0)
    ker = np.ones(smooth) / smooth
    out = np.stack([np.convolve(walk[:, d], ker, mode="same") for d in range(dim)], axis=-1)
    out = out[smooth:smooth + n]
    return out - out[0]


def make_clip(rng, num_frames, num_bodies=24, fps=30.0, speed=1.0):
    """One synthetic clip: dict(local_rot[T,B,4] xyzw, root_trans[T,3], fps, beta, gender, min_verts_h)."""
    t = num_frames
    joint_aa = _smooth_walk(rng, t, 3 * (num_bodies - 1), 0.035 * speed, 6).reshape(t, num_bodies - 1, 3)
    joint_aa = 0.9 * np.tanh(joint_aa / 0.9) + rng.normal(0.0, 0.15, size=(1, num_bodies - 1, 3))
    heading = rng.uniform(-np.pi, np.pi) + _smooth_walk(rng, t, 1, 0.03 * speed, 8)[:, 0]
    wobble = 0.12 * np.tanh(_smooth_walk(rng, t, 3, 0.02 * speed, 6))
    yaw_q = _quat_from_rotvec(np.stack([np.zeros(t), np.zeros(t), heading], axis=-1))
    root_q = _quat_mul(_quat_mul(yaw_q, _quat_from_rotvec(wobble)), np.broadcast_to(BASE_ROT, (t, 4)))
    local = np.concatenate([root_q[:, None, :], _quat_from_rotvec(joint_aa)], axis=1)
    local /= np.linalg.norm(local, axis=-1, keepdims=True)
    xy = _smooth_walk(rng, t, 2, 0.012 * speed, 8)
    z = 0.93 + 0.03 * np.tanh(_smooth_walk(rng, t, 1, 0.02, 8))
    root_trans = np.concatenate([xy, z], axis=-1)
    return {
        "local_rot": local.astype(np.float64),
        "root_trans": root_trans.astype(np.float64),
        "fps": float(fps),
        "beta": rng.normal(0.0, 0.5, size=10),
        "gender": "neutral",
        "min_verts_h": float(rng.uniform(-0.01, 0.03)),
    }


def make_clips(seed, num_clips, min_frames=90, max_frames=300, speed=1.0):
    """`num_clips` clips with lengths U[min_frames, max_frames] (SURVEY.md §8d workloads)."""
    rng = np.random.default_rng(seed)
    lengths = rng.integers(min_frames, max_frames + 1, size=num_clips)
    return [make_clip(rng, int(n), speed=speed) for n in lengths]
