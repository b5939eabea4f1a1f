"""Per-step kernel budget of play_steps from a rocprofv3 kernel trace of tools/play_profile.py:  python tools/play_summary.py DB epochs+1"""
import sqlite3
import sys


def main(db, epochs):
    c = sqlite3.connect(db)
    steps = epochs * 32
    rows = list(c.execute("select name, total_calls, total_duration, average from top_kernels"))
    tot = sum(r[2] for r in rows)
    print("# kernels of %d play_steps epochs (%d steps, warm-up epoch included), per STEP: calls, microseconds (rocprofv3 top_kernels reports microseconds); GPU-busy total %.1f us per step (streams overlap: more than the wall clock)" % (epochs, steps, tot / steps))
    groups = {"physics": 0.0, "gemm": 0.0, "engine (v2p) other": 0.0, "glue (torch elementwise / copies / reductions)": 0.0}
    calls = dict.fromkeys(groups, 0)
    print("%-86s %8s %10s %9s" % ("kernel", "calls", "us", "avg_us"))
    for name, n, total, avg in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:84]
        g = "physics" if "physics_ll" in name else "gemm" if ("Cijk" in name or "gemm" in name.lower()) else "engine (v2p) other" if "v2p::" in name else "glue (torch elementwise / copies / reductions)"
        groups[g] += total / steps
        calls[g] += n / steps
        if total / steps > 1.5:
            print("%-86s %8.2f %10.2f %9.2f" % (short, n / steps, total / steps, avg))
    print()
    for g in groups:
        print("%-50s %7.1f launches per step %9.1f us per step (%.1f %% of GPU-busy)" % (g, calls[g], groups[g], 100 * groups[g] * steps / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
