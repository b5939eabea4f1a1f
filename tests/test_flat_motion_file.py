"""The flat, memory-mappable motion-library file (motion_tables.save_flat / load_flat): round trip, alignment, nothing read until touched."""
import os

import numpy as np
import pytest

from vid2player3d_amd import motion_tables as mt
from vid2player3d_amd import synth
from vid2player3d_amd.model import load_baked_model


@pytest.fixture(scope="module")
def tables():
    bm = load_baked_model()
    return mt.build_tables(synth.make_clips(3, 5, 40, 90), bm.parents, bm.local_pos)


def test_round_trip_mapped_and_copied(tables, tmp_path):
    path = mt.save_flat(str(tmp_path / "lib.v2pm"), tables)
    assert os.path.getsize(path) % 4096 == 0
    for mmap in (True, False):
        got = mt.load_flat(path, mmap=mmap)
        assert set(got) == set(k for k in mt.TABLE_KEYS + mt.CLIP_KEYS if k in tables)
        for k, v in got.items():
            assert v.dtype == np.asarray(tables[k]).dtype and np.array_equal(np.asarray(v), np.asarray(tables[k])), k
            if mmap and v.size:
                assert isinstance(v, np.memmap) and v.offset % 4096 == 0 and not v.flags.writeable


def test_bad_magic_is_refused(tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"not a library" * 10)
    with pytest.raises(ValueError):
        mt.load_flat(str(p))


def test_golden_tables_survive_the_file(tmp_path):
    """the tables recorded from the reference's own MotionLib (tests/golden/motion_tables.npz) through the flat file, bit for bit"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "motion_tables.npz"))
    tabs = {k: g[k] for k in mt.TABLE_KEYS + mt.CLIP_KEYS if k in g.files}
    got = mt.load_flat(mt.save_flat(str(tmp_path / "golden.v2pm"), tabs))
    for k in tabs:
        assert np.array_equal(np.asarray(got[k]), tabs[k]), k


def test_motion_lib_from_the_mapped_file(tables, tmp_path):
    import torch

    from vid2player3d_amd.motion_lib import MotionLib

    path = mt.save_flat(str(tmp_path / "lib.v2pm"), tables)
    a, b = MotionLib(tables, "cpu"), MotionLib.from_flat_file(path, "cpu", single_skeleton=True)
    for k in mt.TABLE_KEYS:
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(a._motion_lengths, b._motion_lengths) and torch.equal(a.length_starts, b.length_starts) and b._single_skeleton
    again = mt.load_flat(b.save_flat(str(tmp_path / "again.v2pm")), mmap=False)
    for k in mt.TABLE_KEYS:
        assert np.array_equal(again[k], np.asarray(tables[k])), k
