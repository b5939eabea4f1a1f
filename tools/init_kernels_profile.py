"""The two init-time kernels of round 5 under rocprofv3: v2p_shapes_compile (2048 body shapes = 49,152 hull jobs of ~290 points) and
v2p_motion_tables_build (2048 clips, ~400 k frames, one skeleton per clip).
    rocprofv3 --kernel-trace --stats -d DIR -o init -- python tools/init_kernels_profile.py ; python tools/rocprof_summary.py DIR/init_results.db"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from vid2player3d_amd import body_shapes as bs, motion_tables, synth  # noqa: E402
from vid2player3d_amd.model import load_baked_model  # noqa: E402

if __name__ == "__main__":
    dev = "cuda:0"
    base = load_baked_model()
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    clouds, rest = bs.deform_clouds(base, bs.family_params(S, seed=7))
    per = [[clouds[b][s] for b in range(24)] for s in range(S)]
    sizes = np.array([[len(c) for c in row] for row in per]).reshape(-1)
    off = np.concatenate([[0], np.cumsum(sizes)])
    pts = np.concatenate([c for row in per for c in row])
    bs.compile_clouds_device(pts[:off[24]], off[:25], dev)  # warm-up
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = bs.compile_clouds_device(pts, off, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("[init] v2p_shapes_compile: %d jobs, %d points (%.1f MB in), %.3f s wall per call (upload + kernel + download)" % (len(sizes), len(pts), pts.nbytes / 1e6, dt))
    clips = synth.make_clips(7, S, 90, 300)
    lp = np.stack([base.local_pos * s for s in np.random.default_rng(0).uniform(0.9, 1.1, size=S)])
    motion_tables.build_tables_device(clips[:4], base.parents, lp[:4], dev)
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tabs = motion_tables.build_tables_device(clips, base.parents, lp, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    F = int(tabs["motion_num_frames"].sum())
    print("[init] v2p_motion_tables_build: %d clips, %d frames: reads %.1f MB (float64 rotations + root translations), writes %.1f MB (float32 tables), %.3f s wall per call "
          "(host concatenation + upload + two kernels)" % (S, F, F * (24 * 4 + 3) * 8 / 1e6, F * 339 * 4 / 1e6, dt))
