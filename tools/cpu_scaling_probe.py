#!/usr/bin/env python
"""How the CPU oracle (bench.py's cpu_baseline leg) scales on this host: cgroup quota, affinity, topology, and env-steps/s of the batched
C physics step at 1 .. all threads (OMP_PLACES / OMP_PROC_BIND as set by the caller).  No GPU involved.
usage: [OMP_PLACES=cores OMP_PROC_BIND=close] python tools/cpu_scaling_probe.py [envs, default 2048]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main(n):
    print("os.cpu_count()", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)), "| OMP_PLACES", os.environ.get("OMP_PLACES"), "| OMP_PROC_BIND", os.environ.get("OMP_PROC_BIND"))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            print(f, "=", open(f).read().strip())
        except OSError:
            pass
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
        print("\n".join(l for l in out.splitlines() if any(k in l for k in ("Model name", "Socket", "Core(s)", "Thread(s)", "NUMA node", "CPU(s):", "MHz"))))
    except OSError:
        pass
    from oracle.phys_oracle import BatchOracle, default_params, lib_fast
    from tools.gain_probe import fixture

    bm, root, dpos, dvel, pd, force, torque = fixture(n, seed=11, lift=0.0, vel_sigma=0.5)
    mx = lib_fast().v2p_oracle_max_threads()
    print("omp max threads", mx)
    ths = sorted(set([1, 2, 4, 8, 16, 32, 64, 128, 256, mx, len(os.sched_getaffinity(0))]))
    for th in [t for t in ths if t <= max(mx, 1)]:
        o = BatchOracle(bm, n, default_params(), threads=th, fast=True)
        o.set_state(root, dpos, dvel)
        o.step(pd, force, torque)  # warm-up
        k, t0 = 0, time.perf_counter()
        while k < 2 or time.perf_counter() - t0 < 2.0:
            o.step(pd, force, torque)
            k += 1
        dt = time.perf_counter() - t0
        print("threads %4d: %9.0f env-steps/s  %7.1f per thread  (%d steps of %d envs)" % (th, n * k / dt, n * k / dt / th, k, n))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2048)
