"""MotionLib: host-side mirror of the reference's reference-motion store
(embodied_pose/utils/motion_lib.py:67-266) over the HIP sampler.

Same attribute names (`gts grs lrs grvs gravs dvs`, `_motion_lengths`, `_motion_num_frames`,
`_motion_dt`, `_motion_fps`, `_motion_weights`, `_motion_bodies`, `_motion_min_verts_h`,
`length_starts`, `motion_ids`) and the same methods (`num_motions`, `get_total_length`,
`get_motion_length`, `sample_motions`, `sample_time`, `get_motion_state`,
`merge_multiple_motion_libs`).  Tables are device-resident torch tensors (288 GB of HBM3E
holds the whole of AMASS at 1356 B/frame); `get_motion_state` is a HIP kernel and has no CPU
path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import motion_tables as mt

KEY_BODY_IDS_DEFAULT = (7, 3, 18, 23)  # R_Ankle, L_Ankle, L_Hand, R_Hand (cfg/amass_im.yaml:17)


class MotionLib:
    def __init__(self, tables, device, key_body_ids=KEY_BODY_IDS_DEFAULT, dof_body_ids=None, dof_offsets=None):
        """`tables`: dict from motion_tables.build_tables (numpy) or of torch tensors."""
        self._device = torch.device(device)
        self._key_body_ids = torch.tensor(list(key_body_ids), device=self._device)
        self._dof_body_ids = list(range(1, 24)) if dof_body_ids is None else list(dof_body_ids)
        self._dof_offsets = list(range(0, 70, 3)) if dof_offsets is None else list(dof_offsets)
        self._num_dof = self._dof_offsets[-1]

        def dev(x, dtype):
            if torch.is_tensor(x):
                t = x
            else:
                import warnings
                with warnings.catch_warnings():  # (a read-only mapped file: the tensor is only ever the source of the copy to the device)
                    warnings.simplefilter("ignore")
                    t = torch.as_tensor(np.asarray(x))
            return t.to(device=self._device, dtype=dtype).contiguous()

        for k in mt.TABLE_KEYS:
            setattr(self, k, dev(tables[k], torch.float32))
        self._motion_lengths = dev(tables["motion_lengths"], torch.float32)
        self._motion_num_frames = dev(tables["motion_num_frames"], torch.int64)
        self._motion_dt = dev(tables["motion_dt"], torch.float32)
        self._motion_fps = dev(tables["motion_fps"], torch.float32)
        self._motion_weights = dev(tables["motion_weights"], torch.float32)
        self._motion_bodies = dev(tables["motion_bodies"], torch.float32)
        self._motion_min_verts_h = dev(tables["motion_min_verts_h"], torch.float32)
        self.generate_length_starts()
        self.motion_ids = torch.arange(len(self._motion_lengths), dtype=torch.long, device=self._device)
        self._handle = None
        self._borrowed = False  # set by the task that hands handle() to an env batch

    # ---- construction helpers -------------------------------------------------------------
    @classmethod
    def from_clips(cls, clips, body_model, device, **kw):
        """Synthetic / converted clips -> tables (motion_tables.py) -> device."""
        # forward kinematics + velocity estimation run on the device (v2p_motion_tables_build); build="host": the numpy statement of it
        # (motion_tables.build_tables, the checker: pinned to the reference's own constructor path)
        build = kw.pop("build", "device" if torch.device(device).type == "cuda" else "host")
        make = (lambda c, p, lp: mt.build_tables_device(c, p, lp, device)) if build == "device" else mt.build_tables
        if isinstance(body_model, (list, tuple)):  # one body shape per clip
            return cls(make(clips, body_model[0].parents, np.stack([m.local_pos for m in body_model])), device, **kw)
        lib = cls(make(clips, body_model.parents, body_model.local_pos), device, **kw)
        lib._single_skeleton = True  # every clip was built on this one skeleton, whatever beta labels the clips carry
        return lib

    @classmethod
    def from_flat_file(cls, path, device, **kw):
        """A library saved with `save_flat` (motion_tables.save_flat): the file is memory-mapped and goes to the device table by table."""
        single = kw.pop("single_skeleton", False)
        lib = cls(mt.load_flat(path, mmap=True), device, **kw)
        lib._single_skeleton = single
        return lib

    def tables(self):
        """the tables as a dict of numpy arrays (what motion_tables.build_tables returns): for save_flat"""
        out = {k: getattr(self, k).cpu().numpy() for k in mt.TABLE_KEYS}
        for k in mt.CLIP_KEYS:
            v = getattr(self, "_" + k, None) if k != "length_starts" else self.length_starts
            if v is not None:
                out[k] = v.cpu().numpy()
        return out

    def save_flat(self, path):
        return mt.save_flat(path, self.tables())

    def generate_length_starts(self):
        """motion_lib.py:95-99"""
        lengths = self._motion_num_frames
        shifted = lengths.roll(1)
        shifted[0] = 0
        self.length_starts = shifted.cumsum(0)

    def merge_multiple_motion_libs(self, motion_lib_arr):
        """motion_lib.py:101-118.  Must happen before the library is handed to a task: an env batch borrows the table pointers."""
        if getattr(self, "_borrowed", False):
            raise RuntimeError("merge_multiple_motion_libs after the library was handed to an env batch: the batch borrows the device tables "
                               "(merge first, then create the task)")
        keys = list(mt.TABLE_KEYS) + ["_motion_weights", "_motion_lengths", "_motion_num_frames", "_motion_dt", "_motion_fps",
                                      "_motion_bodies", "_motion_min_verts_h"]
        for k in keys:
            setattr(self, k, torch.cat([getattr(self, k)] + [getattr(m, k) for m in motion_lib_arr], dim=0).contiguous())
        self.generate_length_starts()
        self._motion_weights = self._motion_weights / self._motion_weights.sum()
        self.motion_ids = torch.arange(len(self._motion_lengths), dtype=torch.long, device=self._device)
        self._release()

    # ---- reference API ---------------------------------------------------------------------
    def num_motions(self):
        return len(self.motion_ids)

    def get_total_length(self):
        return self._motion_lengths.sum()

    def get_motion_length(self, motion_ids):
        return self._motion_lengths[motion_ids]

    def sample_motions(self, n, weights_from_lenth=True):
        """motion_lib.py:129-136"""
        w = self._motion_lengths / self._motion_lengths.sum() if weights_from_lenth else self._motion_weights
        return torch.multinomial(w, num_samples=n, replacement=True)

    def sample_time(self, motion_ids, truncate_time=None, motion_time_range=None):
        """motion_lib.py:138-159"""
        phase = torch.rand(motion_ids.shape, device=self._device)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            motion_len = torch.clamp_min(motion_len - truncate_time, 0)
        if motion_time_range is not None:
            start, end = motion_time_range
            start = torch.ones_like(motion_len) * (0 if start is None else start)
            end = motion_len if end is None else torch.ones_like(motion_len) * end
            return start + (end - start) * phase
        return phase * motion_len

    # ---- the hot path ----------------------------------------------------------------------
    def handle(self):
        """v2p_mlib handle over the device tables (created lazily; tables are borrowed)."""
        if self._handle is None:
            if self._device.type != "cuda":
                raise RuntimeError("MotionLib tables live on %s: the sampler is a HIP kernel and needs a GPU device "
                                   "(there is no CPU fallback)" % self._device)
            lib = _lib.load()
            t = _lib.MotionTables()
            t.num_motions = self.num_motions()
            t.num_frames_total = self.gts.shape[0]
            for k in mt.TABLE_KEYS:
                setattr(t, k, getattr(self, k).data_ptr())
            t.motion_lengths = self._motion_lengths.data_ptr()
            t.motion_num_frames = self._motion_num_frames.data_ptr()
            t.motion_dt = self._motion_dt.data_ptr()
            t.motion_min_verts_h = self._motion_min_verts_h.data_ptr()
            t.length_starts = self.length_starts.data_ptr()
            t.motion_bodies = self._motion_bodies.data_ptr()
            t.key_body_ids[:] = [int(x) for x in self._key_body_ids.tolist()]
            h = C.c_void_p()
            _lib.check(lib.v2p_mlib_create(C.byref(t), self._device.index or 0, C.byref(h)), "v2p_mlib_create")
            self._handle = h
        return self._handle

    def _release(self):
        if self._handle is not None:
            _lib.load().v2p_mlib_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def get_motion_state(self, motion_ids, motion_times, return_rigid_body=False, device=None, adjust_height=False, ground_tolerance=0.0):
        """motion_lib.py:164-266.  Returns (root_pos, root_rot, dof_pos, root_vel, root_ang_vel,
        dof_vel, key_pos[, rb_pos, rb_rot])."""
        lib = _lib.load()
        h = self.handle()
        ids = motion_ids.to(device=self._device, dtype=torch.int64).contiguous().view(-1)
        times = motion_times.to(device=self._device, dtype=torch.float32).contiguous().view(-1)
        q = ids.shape[0]
        shapes = [(q, 3), (q, 4), (q, 69), (q, 3), (q, 3), (q, 69), (q, 4, 3)]
        if return_rigid_body:
            shapes += [(q, 24, 3), (q, 24, 4)]
        outs = [torch.empty(s, dtype=torch.float32, device=self._device) for s in shapes]
        arr = (C.c_void_p * 9)(*([o.data_ptr() for o in outs] + [None] * (9 - len(outs))))
        _lib.check(lib.v2p_motion_state(h, _lib.ptr(ids), _lib.ptr(times), q, int(bool(adjust_height)), float(ground_tolerance),
                                        C.byref(arr), _lib.current_stream(self._device)), "v2p_motion_state")
        if device is not None and torch.device(device) != self._device:
            outs = [o.to(device) for o in outs]
        return tuple(outs)
