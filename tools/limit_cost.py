#!/usr/bin/env python
"""What the joint-limit rows of the racket arm cost, split into "rows exist" and "rows act": bench.py --racket-ball with the player's
ranges (rows often active: R_Wrist_x is +-10 deg), with the same DOFs limited to +-170 deg (rows exist, the walk stops at the two joints,
but they practically never act) and without limit rows.  usage (GPU box): python tools/limit_cost.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CODE = """
import sys
sys.path.insert(0, %r)
from vid2player3d_amd import racket
if %d:
    for lim in racket.PLAYERS["djokovic"]["limits"].values():
        pass
    racket.PLAYERS["djokovic"]["limits"] = {k: tuple((-170.0, 170.0) if r is not None else None for r in v) for k, v in racket.PLAYERS["djokovic"]["limits"].items()}
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--racket-ball", "--steps", "192", "--warmup", "64"] + %r
bench.main()
"""
for name, wide, extra in (("player ranges", 0, []), ("+-170 deg (rows never act)", 1, []), ("no limit rows", 0, ["--joint-limits", "0"])):
    out = subprocess.run([sys.executable, "-c", CODE % (ROOT, wide, extra)], capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
    d = json.loads(out[-1])
    print("%-28s %10.0f env-steps/s  ms/step %.4f  kernel ms %.4f" % (name, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
