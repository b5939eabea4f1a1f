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
    if "GRBM_GUI_ACTIVE" in res and "clock_mhz" not in res:
        # the clock the physics kernel ran at in the GRBM_GUI_ACTIVE pass: busy cycles summed over the 8 XCDs / 8 / the kernel's duration in that pass
        try:
            dur = c.execute("select avg(duration) from kernels where name like '%physics_ll%'").fetchone()[0]
            res["clock_mhz"] = {"avg": res["GRBM_GUI_ACTIVE"]["avg"] / 8.0 / (dur * 1e-3), "kernel_ns_in_that_pass": dur,
                                "how": "GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / kernel duration of the same rocprofv3 pass"}
        except Exception as e:  # (schema of another rocprofv3 version: the counters stay, the clock is left out)
            res["clock_mhz"] = {"error": str(e)}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vid2player3d_amd import build  # noqa: E402

res["kernel_source_sha16"] = build.kernel_source_hash()  # (bench.py quotes these counters only while the kernel's sources still hash to this)
print(json.dumps(res, indent=1))
