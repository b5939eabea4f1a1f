#!/usr/bin/env python
"""Per-launch averages of the SQ counter passes of tools/valu_probe.sh for the physics kernel -> JSON (stdout)."""
import glob
import json
import os
import sqlite3
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
res = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_valu_*", "pmc_results.db"))):
    c = sqlite3.connect(d)
    for cn, n, avg in c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%physics_ll%' group by counter_name"):
        res[cn] = {"launches": n, "avg": avg}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vid2player3d_amd import build  # noqa: E402

res["kernel_source_sha16"] = build.kernel_source_hash()  # (bench.py quotes these counters only while the kernel's sources still hash to this)
print(json.dumps(res, indent=1))
