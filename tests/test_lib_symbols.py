"""CPU checks of the boundary: the C-ABI library builds, loads and exports every symbol include/v2p_rollout.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vid2player3d_amd import build

    path = build.build()
    return ctypes.CDLL(path)


def declared_symbols():
    src = open(os.path.join(REPO, "include", "v2p_rollout.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(v2p_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libv2p_rollout.so does not export %s" % n


def test_binding_covers_the_header():
    from vid2player3d_amd import _lib

    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_abi_version_and_error_string(lib):
    lib.v2p_abi_version.restype = ctypes.c_int
    lib.v2p_last_error.restype = ctypes.c_char_p
    assert lib.v2p_abi_version() == 14
    assert isinstance(lib.v2p_last_error(), bytes)


def test_struct_sizes_match_the_header():
    """ctypes mirrors of the ABI structs must have the C layout (checked against a tiny C program)."""
    import subprocess
    import tempfile

    from vid2player3d_amd import _lib

    prog = r'''
#include <stdio.h>
#include "v2p_rollout.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(v2p_model_desc), sizeof(v2p_motion_tables), sizeof(v2p_sim_cfg), sizeof(v2p_env_buffers), sizeof(v2p_ball_cfg), sizeof(v2p_ball_buffers)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), src, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.ModelDesc), ctypes.sizeof(_lib.MotionTables), ctypes.sizeof(_lib.SimCfg), ctypes.sizeof(_lib.EnvBuffers),
                     ctypes.sizeof(_lib.BallCfg), ctypes.sizeof(_lib.BallBuffers)]


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    lib.v2p_last_error.restype = ctypes.c_char_p
    out = ctypes.c_void_p()
    assert lib.v2p_model_create(None, 0, ctypes.byref(out)) == -1
    assert b"null" in lib.v2p_last_error()
    assert lib.v2p_env_step(None, None, None) == -1


def test_no_cpu_fallback():
    """The product path must fail loudly without the HIP engine: CPU device types are refused."""
    from vid2player3d_amd.tasks import HumanoidSMPLIM, default_cfg

    with pytest.raises(RuntimeError):
        HumanoidSMPLIM(default_cfg(4), device_type="cpu", device_id=0)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "vid2player3d_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", txt, flags=re.M), f
                assert not re.search(r"#include[^\n]*oracle", txt), f
                assert "libv2p_phys_oracle" not in txt and "task_oracle" not in txt, f


def test_model_blob_and_table_builder(golden_tables):
    from vid2player3d_amd import motion_tables, synth
    from vid2player3d_amd.model import load_baked_model

    m = load_baked_model()
    assert m.num_bodies == 24 and m.num_dof == 69 and abs(m.total_mass - 102.418) < 1e-2
    assert list(m.parents) == [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 17, 11, 19, 20, 21, 22]
    clips = synth.make_clips(seed=3, num_clips=3, min_frames=34, max_frames=60)
    tabs = motion_tables.build_tables(clips, m.parents, m.local_pos)
    for k in motion_tables.TABLE_KEYS:
        assert np.abs(tabs[k].astype(np.float64) - golden_tables[k]).max() < 1e-6, k
    for k in ("motion_lengths", "motion_num_frames", "motion_dt", "motion_bodies", "motion_min_verts_h", "length_starts"):
        assert np.array_equal(tabs[k], golden_tables[k]), k


def test_new_entry_points_refuse_bad_arguments_without_touching_a_gpu():
    """Argument validation of the ABI 13 entry points happens before any HIP call: bad sizes / null buffers come back as V2P_ERR_INVALID
    with a message (no compute call is made here)."""
    from vid2player3d_amd import _lib

    L = _lib.load()
    INVALID = -1
    null = None
    one = ctypes.c_void_p(16)  # (a non-null placeholder: every call below is refused before anything is dereferenced)
    assert L.v2p_shapes_compile(-1, null, null, 0, null, null, 1, 900.0, 64, 1e-10, null, null, null, null, null, null, null, null) == INVALID
    assert L.v2p_shapes_compile(4, one, one, 100, one, one, 1, 900.0, 65, 1e-10, one, one, one, one, one, one, one, null) == INVALID  # max_verts > 64
    assert L.v2p_shapes_compile(4, null, one, 100, one, one, 1, 900.0, 64, 1e-10, one, one, one, one, one, one, one, null) == INVALID  # null points
    assert b"v2p_shapes_compile" in L.v2p_last_error()
    par = (ctypes.c_int32 * 24)(*([-1] + list(range(23))))
    assert L.v2p_motion_tables_build(-1, 1, null, null, null, null, null, null, par, null, 0, null, null, null, null, null, null, null) == INVALID
    assert L.v2p_motion_tables_build(10, 1, null, one, one, one, one, one, par, one, 0, one, one, one, one, one, one, null) == INVALID
    assert L.v2p_motion_tables_build(10, 1, one, one, one, one, one, one, None, one, 0, one, one, one, one, one, one, null) == INVALID  # no tree
    assert L.v2p_rollout_record(-1, null, 461, *([null] * 15)) == INVALID
    assert L.v2p_rollout_record(8, null, 461, one, one, one, one, one, one, one, one, one, one, one, one, one, one, null) == INVALID  # rows to copy, no obs
    assert L.v2p_value_record(8, null, null, null, 1e-5, null, null, null, null) == INVALID
    assert L.v2p_value_record(8, one, one, null, 1e-5, null, one, null, null) == INVALID      # mean without var
    assert L.v2p_value_record(8, one, null, null, 1e-5, null, null, one, null) == INVALID     # next_values without terminated
    assert L.v2p_policy_head_record(8, one, one, 48, 48, one, one, one, one, one, null, null, null) == INVALID  # frame outside the context window
    assert L.v2p_policy_head_record(8, one, one, 48, 8, one, one, one, one, null, null, null, null) == INVALID   # no neglogp row
    # zero-sized calls are no-ops, not errors
    assert L.v2p_rollout_record(0, null, 461, *([null] * 15)) == 0 and L.v2p_value_record(0, null, null, null, 1e-5, null, null, null, null) == 0
    assert L.v2p_shapes_compile(0, null, null, 0, null, null, 1, 900.0, 64, 1e-10, null, null, null, null, null, null, null, null) == 0
