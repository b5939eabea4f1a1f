#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) into a short text table.

    python tools/rocprof_summary.py gpurun_out/prof/stats_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats summary of %s (durations in microseconds)" % path)
    print("%-72s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if len(short) > 70:
            short = short[:67] + "..."
        print("%-72s %8d %14.1f %12.3f %8.3f" % (short, calls, total, avg, pct))
    cur = c.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x, count(*), avg(duration), min(duration), max(duration) "
                    "from kernels where name like '%v2p::%' group by name")
    print("\n# per-dispatch resources of the engine's kernels (duration ns)")
    print("%-44s %5s %5s %5s %7s %7s %9s %5s %6s %12s %12s %12s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid", "wg", "n", "avg_ns", "min_ns", "max_ns"))
    for r in cur:
        short = r[0].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        print("%-44s %5d %5d %5d %7d %7d %9d %5d %6d %12.0f %12.0f %12.0f" % ((short[:44],) + tuple(r[1:])))


if __name__ == "__main__":
    main(sys.argv[1])
