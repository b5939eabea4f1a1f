"""Host-side pieces of the per-clip body shapes: uniformly scaled models and per-clip skeleton tables."""
import numpy as np

from vid2player3d_amd import motion_tables, synth
from vid2player3d_amd.model import load_baked_model


def test_scaled_model_follows_the_scaling_laws():
    base = load_baked_model()
    s = 1.2
    m = base.scaled(s)
    assert np.allclose(m.local_pos, base.local_pos * s) and np.allclose(m.com, base.com * s)
    assert np.allclose(m.hull_verts, base.hull_verts * s) and np.array_equal(m.hull_offsets, base.hull_offsets)
    assert np.allclose(m.mass, base.mass * s ** 3) and np.allclose(m.inertia, base.inertia * s ** 5)
    assert abs(m.total_mass - base.total_mass * s ** 3) < 1e-9
    # gains scale with total mass / 90 like every asset's (humanoid_smpl_im.py:376-385)
    assert np.allclose(m.kp, base.kp * s ** 3) and np.allclose(m.kd, base.kd * s ** 3)
    assert np.array_equal(m.parents, base.parents) and m.body_names == base.body_names


def test_tables_with_one_skeleton_per_clip():
    base = load_baked_model()
    shapes = [base.scaled(0.9), base, base.scaled(1.1)]
    clips = synth.make_clips(4, 3, 30, 40)
    one = motion_tables.build_tables(clips, base.parents, base.local_pos)
    per = motion_tables.build_tables(clips, base.parents, np.stack([m.local_pos for m in shapes]))
    nf = one["motion_num_frames"]
    a, b = 0, int(nf[0])
    c = b + int(nf[1])
    # clip 1 uses the unscaled skeleton: identical rows; clips 0 and 2: body offsets from the root scale with the skeleton
    assert np.array_equal(per["gts"][b:c], one["gts"][b:c]) and np.array_equal(per["grs"], one["grs"])
    rel_one = one["gts"][a:b, 1:] - one["gts"][a:b, :1]
    rel_per = per["gts"][a:b, 1:] - per["gts"][a:b, :1]
    assert np.allclose(rel_per, 0.9 * rel_one, atol=2e-6)
    rel_one = one["gts"][c:, 1:] - one["gts"][c:, :1]
    rel_per = per["gts"][c:, 1:] - per["gts"][c:, :1]
    assert np.allclose(rel_per, 1.1 * rel_one, atol=2e-6)
