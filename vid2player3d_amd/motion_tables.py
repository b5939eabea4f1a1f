"""Host-side builder of the flat reference-motion tables (numpy, init-time only).

Mirrors what the reference does offline, so that synthetic clips can be turned into motion
tables without poselib:

  * forward kinematics over the 24-node tree  -- poselib/poselib/skeleton/skeleton3d.py:409-431
    (`global_transformation`; `transform_mul` in poselib/poselib/core/rotation3d.py:325-334
    re-normalises every composed quaternion to unit length with w >= 0)
  * finite-difference + Gaussian(sigma=2) linear / angular velocities -- skeleton3d.py:1226-1249
  * joint-frame dof velocities from consecutive local rotations
    -- embodied_pose/utils/motion_lib.py:443-458, 490-519
  * concatenation into gts/grs/lrs/grvs/gravs/dvs + per-clip vectors -- motion_lib.py:78-99, 370-384

The hot path (`get_motion_state`) never runs here; it is a HIP kernel (csrc/motion_state.hip).
"""
import numpy as np

from . import synth

TABLE_KEYS = ("gts", "grs", "lrs", "grvs", "gravs", "dvs")
CLIP_KEYS = ("motion_lengths", "motion_num_frames", "motion_dt", "motion_fps", "motion_weights",
             "motion_bodies", "motion_min_verts_h", "length_starts")

_GENDER_ID = {"neutral": 0.0, "male": 1.0, "female": 2.0}


def quat_mul(a, b):
    return synth._quat_mul(a, b)


def quat_conj(q):
    return np.concatenate([-q[..., :3], q[..., 3:]], axis=-1)


def quat_normalize_pos(q):
    """poselib `quat_normalize`: flip to w >= 0, then unit length (rotation3d.py:95-101)."""
    q = np.where(q[..., 3:] < 0, -q, q)
    return q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-9)


def quat_rotate(q, v):
    """poselib `quat_rotate` (rotation3d.py:208-214): q * (v,0) * conj(q)."""
    vq = np.concatenate([v, np.zeros_like(v[..., :1])], axis=-1)
    return quat_mul(quat_mul(q, vq), quat_conj(q))[..., :3]


def quat_angle_axis(q):
    """poselib `quat_angle_axis` (rotation3d.py:233-242): angle in [0, pi] from 2w^2-1."""
    one = q.dtype.type(1.0)
    s = q.dtype.type(2.0) * q[..., 3] ** 2 - one
    angle = np.arccos(np.clip(s, -one, one))
    axis = q[..., :3] / np.maximum(np.linalg.norm(q[..., :3], axis=-1, keepdims=True), q.dtype.type(1e-9))
    return angle, axis


def forward_kinematics(local_rot, root_trans, parents, local_pos):
    """Global rotation [T,B,4] and translation [T,B,3] from local rotations + root translation."""
    t, b = local_rot.shape[:2]
    grot = np.zeros((t, b, 4))
    gpos = np.zeros((t, b, 3))
    for j in range(b):
        p = int(parents[j])
        if p < 0:
            grot[:, j] = local_rot[:, j]
            gpos[:, j] = root_trans
        else:
            grot[:, j] = quat_normalize_pos(quat_mul(grot[:, p], local_rot[:, j]))
            gpos[:, j] = quat_rotate(grot[:, p], np.broadcast_to(local_pos[j], (t, 3))) + gpos[:, p]
    return grot, gpos


def _gaussian_filter_time(x, sigma=2):
    from scipy.ndimage import gaussian_filter1d

    return gaussian_filter1d(x, sigma, axis=0, mode="nearest")


def clip_to_tables(clip, parents, local_pos):
    """One clip -> dict of per-frame tables (float64)."""
    lrot = np.asarray(clip["local_rot"], dtype=np.float64)
    fps = float(clip["fps"])
    dt = 1.0 / fps
    t = lrot.shape[0]
    grot, gpos = forward_kinematics(lrot, np.asarray(clip["root_trans"], dtype=np.float64), parents, local_pos)
    # linear velocity: central differences then Gaussian smoothing (skeleton3d.py:1226-1233)
    gvel = _gaussian_filter_time(np.gradient(gpos, axis=0)) / dt
    # angular velocity from consecutive global rotations (skeleton3d.py:1236-1249)
    # NB: poselib allocates the difference quaternion with `quat_identity_like` (float32) and
    # assigns the float64 product into it, so angle/axis are evaluated in float32; mirrored
    # here because arccos near 1 makes that rounding visible (~1e-3 rad/s) in `gravs`.
    diff = np.zeros(grot.shape, dtype=np.float32)
    diff[..., 3] = 1.0
    diff[:-1] = quat_normalize_pos(quat_mul(grot[1:], quat_conj(grot[:-1]))).astype(np.float32)
    ang, axis = quat_angle_axis(diff)
    gangvel = _gaussian_filter_time((axis * ang[..., None] / np.float32(dt)).astype(np.float32))
    # dof velocities, expressed in the frame of the earlier pose (motion_lib.py:490-519)
    dq = quat_normalize_pos(quat_mul(quat_conj(lrot[:-1]), lrot[1:]))
    dang, daxis = quat_angle_axis(dq)
    lvel = daxis * dang[..., None] / dt
    dvs = lvel[:, 1:, :].reshape(t - 1, -1)
    dvs = np.concatenate([dvs, dvs[-1:]], axis=0)  # last frame repeats (motion_lib.py:455)
    return {
        "gts": gpos, "grs": grot, "lrs": lrot,
        "grvs": gvel[:, 0], "gravs": gangvel[:, 0], "dvs": dvs,
        "fps": fps, "dt": dt, "num_frames": t, "length": dt * (t - 1),
    }


def build_tables(clips, parents, local_pos):
    """All clips -> flat float32 tables + per-clip vectors (motion_lib.py:78-99, 370-384)."""
    # local_pos [24,3] for one skeleton, or [C,24,3]: one skeleton per clip (each clip of the reference carries its own SMPL shape)
    lp = np.asarray(local_pos)
    per = [clip_to_tables(c, parents, lp[i] if lp.ndim == 3 else lp) for i, c in enumerate(clips)]
    out = {k: np.concatenate([p[k] for p in per], axis=0).astype(np.float32) for k in TABLE_KEYS}
    nf = np.array([p["num_frames"] for p in per], dtype=np.int64)
    out["motion_num_frames"] = nf
    out["motion_lengths"] = np.array([p["length"] for p in per], dtype=np.float32)
    out["motion_dt"] = np.array([p["dt"] for p in per], dtype=np.float32)
    out["motion_fps"] = np.array([p["fps"] for p in per], dtype=np.float32)
    w = np.full(len(per), 1.0 / len(per), dtype=np.float32)
    out["motion_weights"] = w / w.sum()
    out["motion_bodies"] = np.stack([
        np.concatenate([[_GENDER_ID[c["gender"]]], np.asarray(c["beta"], dtype=np.float64)]) for c in clips
    ]).astype(np.float32)
    out["motion_min_verts_h"] = np.array([c["min_verts_h"] for c in clips], dtype=np.float32)
    starts = np.roll(nf, 1)
    starts[0] = 0
    out["length_starts"] = np.cumsum(starts).astype(np.int64)
    return out


def build_tables_device(clips, parents, local_pos, device):
    """build_tables on the HIP engine (v2p_motion_tables_build: forward kinematics + velocity estimation of all frames of all clips in
    two launches; build_tables above is the numpy statement of the same computation and its checker): the frame tables come back as
    float32 torch tensors on `device`, the per-clip vectors as numpy arrays - what MotionLib() takes.  No CPU path."""
    import ctypes as C

    import torch

    from . import _lib

    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("build_tables_device runs on the HIP engine only (device=%r)" % (device,))
    nf = np.array([len(c["local_rot"]) for c in clips], dtype=np.int64)
    if nf.min() < 2:
        raise ValueError("every clip needs at least 2 frames")
    F, Cn = int(nf.sum()), len(clips)
    starts = np.concatenate([[0], np.cumsum(nf)[:-1]]).astype(np.int64)
    fps = np.array([float(c["fps"]) for c in clips], dtype=np.float64)
    dt = 1.0 / fps
    lp = np.asarray(local_pos, dtype=np.float64)
    per_clip = lp.ndim == 3
    if lp.shape[-2:] != (24, 3) or (per_clip and lp.shape[0] != Cn):
        raise ValueError("local_pos must be [24,3] or [num_clips,24,3]")
    par = np.ascontiguousarray(parents, dtype=np.int32)
    with torch.cuda.device(dev):
        t = lambda a, d: torch.as_tensor(np.ascontiguousarray(a), dtype=d).to(dev)  # noqa: E731
        lrot = t(np.concatenate([np.asarray(c["local_rot"], dtype=np.float64) for c in clips], axis=0), torch.float64)
        rtr = t(np.concatenate([np.asarray(c["root_trans"], dtype=np.float64) for c in clips], axis=0), torch.float64)
        fclip = t(np.repeat(np.arange(Cn, dtype=np.int32), nf), torch.int32)
        d_start, d_nf, d_dt, d_lp = t(starts, torch.int64), t(nf, torch.int32), t(dt, torch.float64), t(lp, torch.float64)
        out = {"gts": torch.empty((F, 24, 3), dtype=torch.float32, device=dev), "grs": torch.empty((F, 24, 4), dtype=torch.float32, device=dev),
               "lrs": torch.empty((F, 24, 4), dtype=torch.float32, device=dev), "grvs": torch.empty((F, 3), dtype=torch.float32, device=dev),
               "gravs": torch.empty((F, 3), dtype=torch.float32, device=dev), "dvs": torch.empty((F, 69), dtype=torch.float32, device=dev)}
        _lib.check(lib.v2p_motion_tables_build(F, Cn, _lib.ptr(lrot), _lib.ptr(rtr), _lib.ptr(fclip), _lib.ptr(d_start), _lib.ptr(d_nf), _lib.ptr(d_dt),
                                               par.ctypes.data_as(_lib.c_i32), _lib.ptr(d_lp), int(per_clip), *[_lib.ptr(out[k]) for k in TABLE_KEYS],
                                               _lib.current_stream(dev)), "v2p_motion_tables_build")
        torch.cuda.current_stream(dev).synchronize()  # (the float64 inputs go out of scope here)
    out["motion_num_frames"] = nf
    out["motion_lengths"] = (dt * (nf - 1)).astype(np.float32)
    out["motion_dt"] = dt.astype(np.float32)
    out["motion_fps"] = fps.astype(np.float32)
    w = np.full(Cn, 1.0 / Cn, dtype=np.float32)
    out["motion_weights"] = w / w.sum()
    out["motion_bodies"] = np.stack([np.concatenate([[_GENDER_ID[c["gender"]]], np.asarray(c["beta"], dtype=np.float64)]) for c in clips]).astype(np.float32)
    out["motion_min_verts_h"] = np.array([c["min_verts_h"] for c in clips], dtype=np.float32)
    out["length_starts"] = starts
    return out


# ---- one flat, memory-mappable file per motion library -----------------------------------------------------------------------------------
# (an AMASS-sized library is gigabytes of frame tables: the reference unpickles all of it into host memory, then copies it to the device,
# `utils/motion_lib.py:67-135`; a mapped file is paged in once, straight into the host-to-device copy, and shared between the ranks of a node)
FLAT_MAGIC = b"V2PMLIB1"
_FLAT_ALIGN = 4096


def save_flat(path, tables):
    """tables: the dict of build_tables (or MotionLib.tables()).  Layout: magic 8 B | header length u64 | JSON header {key: [dtype, shape,
    offset]} | arrays, each at a 4096-byte boundary, C order, little endian."""
    import json

    keys = [k for k in TABLE_KEYS + CLIP_KEYS if k in tables]
    arrs = {k: np.ascontiguousarray(np.asarray(tables[k])) for k in keys}
    header, off = {}, 0
    for k in keys:
        header[k] = [arrs[k].dtype.str, list(arrs[k].shape), off]
        off += -(-arrs[k].nbytes // _FLAT_ALIGN) * _FLAT_ALIGN
    blob = json.dumps(header).encode()
    data0 = -(-(16 + len(blob)) // _FLAT_ALIGN) * _FLAT_ALIGN
    with open(path, "wb") as f:
        f.write(FLAT_MAGIC)
        f.write(np.uint64(len(blob)).tobytes())
        f.write(blob)
        for k in keys:
            f.seek(data0 + header[k][2])
            f.write(arrs[k].tobytes())
        f.truncate(data0 + off)
    return path


def load_flat(path, mmap=True):
    """-> dict of arrays; mmap=True: read-only views of the mapped file (np.memmap), nothing is read until it is touched."""
    import json

    with open(path, "rb") as f:
        if f.read(8) != FLAT_MAGIC:
            raise ValueError("%s is not a flat motion library (bad magic)" % path)
        n = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
        header = json.loads(f.read(n).decode())
    data0 = -(-(16 + n) // _FLAT_ALIGN) * _FLAT_ALIGN
    out = {}
    for k, (dt, shape, off) in header.items():
        if mmap:
            out[k] = np.memmap(path, dtype=np.dtype(dt), mode="r", offset=data0 + off, shape=tuple(shape)) if int(np.prod(shape)) else np.zeros(shape, dtype=np.dtype(dt))
        else:
            with open(path, "rb") as f:
                f.seek(data0 + off)
                out[k] = np.frombuffer(f.read(int(np.prod(shape)) * np.dtype(dt).itemsize), dtype=np.dtype(dt)).reshape(shape).copy()
    return out
