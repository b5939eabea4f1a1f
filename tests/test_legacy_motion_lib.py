"""The reference's pickled motion libraries (uhc/utils/convert_amass_isaac.py:168-176) load without the reference's classes, and
nothing but that one class + tensors is allowed through the unpickler.  Fixtures were written by the reference's own MotionLib
class (oracle/gen_golden_legacy_pth.py)."""
import io
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import task_oracle as O
from tests.conftest import GOLDEN
from vid2player3d_amd.legacy_motion_lib import load_legacy_motion_lib, read_legacy_record, save_tables_npz

PARTS = os.path.join(GOLDEN, "legacy_mlib")


@pytest.fixture(scope="module")
def expected():
    with np.load(os.path.join(GOLDEN, "legacy_mlib_expected.npz")) as z:
        return {k: z[k] for k in z.files}


def test_directory_of_parts_merges_like_the_reference(expected):
    lib = load_legacy_motion_lib(PARTS, "cpu")
    for k in ("gts", "grs", "lrs", "grvs", "gravs", "dvs", "_motion_weights", "_motion_lengths", "_motion_num_frames", "_motion_dt", "_motion_fps",
              "_motion_bodies", "_motion_min_verts_h", "length_starts", "motion_ids"):
        assert np.array_equal(getattr(lib, k).numpy(), expected[k]), k
    assert lib.num_motions() == 3 and len(lib._motion_seq_names) == 3
    assert np.array_equal(lib._motion_body_scales.numpy(), expected["_motion_body_scales"])
    # the numpy oracle on the loaded tables reproduces the reference's get_motion_state on the merged library
    tabs = {k: getattr(lib, k).numpy() for k in ("gts", "grs", "lrs", "grvs", "gravs", "dvs")}
    tabs.update(motion_lengths=lib._motion_lengths.numpy(), motion_num_frames=lib._motion_num_frames.numpy(), motion_dt=lib._motion_dt.numpy(),
                motion_min_verts_h=lib._motion_min_verts_h.numpy(), length_starts=lib.length_starts.numpy())
    res = O.get_motion_state(tabs, expected["state_ids"], expected["state_times"], True, 0.0)
    for name, r in zip(O.MOTION_STATE_NAMES, res):
        err = np.abs(r - expected["state_" + name]).max()
        assert err < 5e-6, (name, err)


def test_single_file_range_and_flat_export(tmp_path, expected):
    one = load_legacy_motion_lib(os.path.join(PARTS, "mlib_part_00001.pth"), "cpu")
    assert one.num_motions() == 1
    sub = load_legacy_motion_lib(PARTS, "cpu", motion_file_range=[1, 2])
    assert torch.equal(sub.gts, one.gts)
    n0 = int(expected["_motion_num_frames"][:2].sum())
    assert np.array_equal(one.gts.numpy(), expected["gts"][n0:])
    out = tmp_path / "flat.npz"
    save_tables_npz(load_legacy_motion_lib(PARTS, "cpu"), str(out))
    with np.load(str(out), allow_pickle=False) as z:
        assert np.array_equal(z["gts"], expected["gts"]) and np.array_equal(z["motion_weights"], expected["_motion_weights"])


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo pwned",))


def test_unpickler_refuses_everything_else(tmp_path):
    bad = tmp_path / "bad.pth"
    torch.save({"x": _Evil()}, str(bad))
    with pytest.raises(pickle.UnpicklingError):
        read_legacy_record(str(bad))
    plain = tmp_path / "plain.pth"
    torch.save({"gts": torch.zeros(3)}, str(plain))
    with pytest.raises(ValueError):
        read_legacy_record(str(plain))  # loads (tensors are fine) but is not a MotionLib
