"""Reference-motion tables built on the device (row g-1: poselib forward kinematics + velocity estimation, csrc/motion_build.hip,
v2p_motion_tables_build) against (a) tests/golden/motion_tables.npz - tables the REFERENCE's own constructor path (poselib SkeletonState /
SkeletonMotion + MotionLib) produced from the same seeded clips (oracle/gen_golden.py) - and (b) the numpy statement of the computation
(motion_tables.build_tables, itself pinned to (a))."""
import time

import numpy as np
import pytest
import torch

from vid2player3d_amd import motion_tables, synth
from vid2player3d_amd.model import load_baked_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# float32 tables from float64 arithmetic on both sides: a value may round to the neighbouring float32.  `gravs` is evaluated in float32
# by poselib itself (acos next to 1: the reference's own noise there is ~1e-3 rad/s, see motion_tables.clip_to_tables)
TOL = {"gts": 2e-6, "grs": 2e-6, "lrs": 2e-7, "grvs": 5e-6, "gravs": 5e-5, "dvs": 2e-5}


def _np(x):
    return x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def test_device_tables_match_the_references_constructor_path(golden_tables):
    m = load_baked_model()
    clips = synth.make_clips(seed=3, num_clips=3, min_frames=34, max_frames=60)  # the clips the golden file was generated from
    tabs = motion_tables.build_tables_device(clips, m.parents, m.local_pos, DEV)
    for k in motion_tables.TABLE_KEYS:
        err = np.abs(_np(tabs[k]).astype(np.float64) - golden_tables[k]).max()
        print("[motion build] %-6s vs reference golden: max err %.2e" % (k, err))
        assert err < TOL[k], (k, err)
    for k in ("motion_lengths", "motion_num_frames", "motion_dt", "motion_bodies", "motion_min_verts_h", "length_starts"):
        assert np.array_equal(_np(tabs[k]), golden_tables[k]), k


@pytest.mark.parametrize("per_clip", [False, True])
def test_device_tables_match_numpy_on_ragged_clips(per_clip):
    """64 clips of 2 ... 300 frames (the 17-tap filter is wider than the short ones: edge replication on both sides at once), one skeleton or
    one per clip (per-clip body shapes)."""
    m = load_baked_model()
    rng = np.random.default_rng(4)
    lens = [2, 3, 5, 9, 16, 17, 18] + list(rng.integers(20, 301, size=57))
    clips = [synth.make_clip(rng, int(n), speed=float(rng.uniform(0.5, 2.5))) for n in lens]
    lp = np.stack([m.local_pos * s for s in rng.uniform(0.8, 1.2, size=len(clips))]) if per_clip else m.local_pos
    want = motion_tables.build_tables(clips, m.parents, lp)
    got = motion_tables.build_tables_device(clips, m.parents, lp, DEV)
    for k in motion_tables.TABLE_KEYS:
        a, b = _np(got[k]).astype(np.float64), want[k].astype(np.float64)
        assert a.shape == b.shape, k
        err = np.abs(a - b)
        print("[motion build] %-6s per_clip=%d: max err %.2e (max |x| %.2f), exact in %.4f of the entries" % (k, per_clip, err.max(), np.abs(b).max(), (err == 0).mean()))
        assert err.max() < TOL[k] * max(1.0, np.abs(b).max() / 10.0), (k, err.max())
    for k in motion_tables.CLIP_KEYS:
        assert np.array_equal(_np(got[k]), want[k]), k


def test_library_built_on_the_device_samples_like_the_host_built_one():
    """MotionLib.from_clips builds on the device by default; get_motion_state over both libraries agrees to the tables' tolerance."""
    from vid2player3d_amd.motion_lib import MotionLib

    m = load_baked_model()
    clips = synth.make_clips(11, 16, 40, 120)
    dev_lib = MotionLib.from_clips(clips, m, DEV)
    host_lib = MotionLib.from_clips(clips, m, DEV, build="host")
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    ids = torch.randint(0, 16, (4000,), device=DEV, generator=g)
    times = torch.rand(4000, device=DEV, generator=g) * dev_lib._motion_lengths[ids] * 1.1 - 0.05
    for a, b, name in zip(dev_lib.get_motion_state(ids, times, return_rigid_body=True), host_lib.get_motion_state(ids, times, return_rigid_body=True),
                          ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos", "rb_pos", "rb_rot")):
        err = (a - b).abs().max().item()
        assert err < (1e-4 if name in ("root_ang_vel", "dof_vel") else 5e-6), (name, err)


def test_amass_sized_library_builds_in_seconds():
    """2048 clips x 90..300 frames (~400 k frames, 0.5 GB of tables), one skeleton per clip: ~20 s of numpy; seconds here (clip synthesis excluded)."""
    m = load_baked_model()
    clips = synth.make_clips(7, 2048, 90, 300)
    lp = np.stack([m.local_pos * s for s in np.random.default_rng(0).uniform(0.9, 1.1, size=2048)])
    motion_tables.build_tables_device(clips[:4], m.parents, lp[:4], DEV)
    t0 = time.perf_counter()
    tabs = motion_tables.build_tables_device(clips, m.parents, lp, DEV)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    F = int(tabs["motion_num_frames"].sum())
    print("[motion build] 2048 clips, %d frames: %.2f s (host concatenation + upload + two launches)" % (F, dt))
    assert dt < 15.0 and tabs["gts"].shape == (F, 24, 3)
    k = 1234
    one = motion_tables.build_tables([clips[k]], m.parents, lp[k])
    s0 = int(tabs["length_starts"][k])
    for key in motion_tables.TABLE_KEYS:
        a = _np(tabs[key][s0:s0 + len(one[key])]).astype(np.float64)
        assert np.abs(a - one[key]).max() < TOL[key] * max(1.0, np.abs(one[key]).max() / 10.0), key
