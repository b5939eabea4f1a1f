"""Loader for the reference's on-disk motion libraries: pickled `utils.motion_lib.MotionLib` objects written by
`uhc/utils/convert_amass_isaac.py:168-176` (`torch.save(motion_lib, "mlib_part_%05d.pth")`) and read back by
`HumanoidSMPLIM._load_motion` (`humanoid_smpl_im.py:420-440`: one file, or a directory whose parts are merged into the first).

The reference needs its own class importable to unpickle them.  Here a restricted unpickler maps that one class onto a plain
record, lets tensors / OrderedDict through and refuses everything else (a .pth from the internet cannot run code), then hands
the tables to this package's `MotionLib`.  `save_tables_npz` writes the flat, mmap-able form the engine prefers.
"""
import glob
import os
import pickle
import types

import numpy as np
import torch

from .motion_lib import MotionLib

_TENSOR_ATTRS = ("gts", "grs", "lrs", "grvs", "gravs", "dvs")
_CLIP_ATTRS = {"_motion_lengths": "motion_lengths", "_motion_num_frames": "motion_num_frames", "_motion_dt": "motion_dt", "_motion_fps": "motion_fps",
               "_motion_weights": "motion_weights", "_motion_bodies": "motion_bodies", "_motion_min_verts_h": "motion_min_verts_h"}
_OPTIONAL = ("_motion_body_scales", "_motion_body_idx", "_motion_seq_ids", "_motion_seq_names")

_ALLOWED = {
    ("collections", "OrderedDict"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
    ("torch", "FloatStorage"), ("torch", "DoubleStorage"), ("torch", "HalfStorage"), ("torch", "LongStorage"), ("torch", "IntStorage"),
    ("torch", "ShortStorage"), ("torch", "CharStorage"), ("torch", "ByteStorage"), ("torch", "BoolStorage"), ("torch.storage", "UntypedStorage"),
    ("torch", "device"), ("torch", "Size"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
}


class LegacyMotionLibRecord:
    """What a pickled reference MotionLib turns into: its attribute dict, nothing else."""


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == "MotionLib" and module.split(".")[-1] == "motion_lib":
            return LegacyMotionLibRecord
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("legacy motion lib: refusing to unpickle %s.%s" % (module, name))


_restricted_pickle = types.SimpleNamespace(Unpickler=_RestrictedUnpickler, load=lambda f, **kw: _RestrictedUnpickler(f, **kw).load(),
                                           __name__="v2p_restricted_pickle")


def read_legacy_record(path):
    rec = torch.load(path, map_location="cpu", pickle_module=_restricted_pickle, weights_only=False)
    if not isinstance(rec, LegacyMotionLibRecord):
        raise ValueError("%s does not hold a pickled MotionLib" % path)
    missing = [k for k in _TENSOR_ATTRS + tuple(_CLIP_ATTRS) if not hasattr(rec, k)]
    if missing:
        raise ValueError("%s: pickled MotionLib lacks %s" % (path, ", ".join(missing)))
    return rec


def record_to_tables(rec):
    """-> the flat table dict `MotionLib(tables, device)` takes (motion_tables.TABLE_KEYS + per-clip vectors)."""
    npy = lambda x: x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)  # noqa: E731
    t = {k: npy(getattr(rec, k)).astype(np.float32) for k in _TENSOR_ATTRS}
    for src, dst in _CLIP_ATTRS.items():
        t[dst] = npy(getattr(rec, src))
    t["motion_num_frames"] = t["motion_num_frames"].astype(np.int64)
    return t


def load_legacy_motion_lib(motion_file, device, motion_file_range=None):
    """`HumanoidSMPLIM._load_motion` (humanoid_smpl_im.py:420-440): a .pth file, or a directory of parts (sorted, optionally sliced by
    `motion_file_range`) merged into the first one with the reference's `merge_multiple_motion_libs` semantics (weights renormalised
    over the concatenation)."""
    if os.path.isdir(motion_file):
        files = sorted(glob.glob(os.path.join(motion_file, "*.pth")))
        if motion_file_range is not None:
            files = files[motion_file_range[0]:motion_file_range[1]]
    else:
        files = [motion_file]
    if not files:
        raise FileNotFoundError("no .pth motion libraries under %s" % motion_file)
    recs = [read_legacy_record(f) for f in files]
    libs = [MotionLib(record_to_tables(r), device) for r in recs]
    lib = libs[0]
    if len(libs) > 1:
        lib.merge_multiple_motion_libs(libs[1:])
    for k in _OPTIONAL:  # kept for callers that look at them (the engine itself does not)
        vals = [getattr(r, k, None) for r in recs]
        if all(v is not None for v in vals):
            setattr(lib, k, sum((list(v) for v in vals), []) if isinstance(vals[0], list) else torch.cat([torch.as_tensor(v) for v in vals], dim=0))
    lib.motion_lib_files = files
    return lib


def save_tables_npz(lib, path):
    """Flat form of a motion library (what `cfg['env']['motion_file']` = *.npz loads): plain arrays, no pickled classes."""
    out = {k: getattr(lib, k).detach().cpu().numpy() for k in _TENSOR_ATTRS}
    for src, dst in _CLIP_ATTRS.items():
        out[dst] = getattr(lib, src).detach().cpu().numpy()
    np.savez(path, **out)
