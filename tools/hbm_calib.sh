#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/ubench/hbm_calib.hip) in separate rocprofv3 passes -> gpurun_out/hbm_calib.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_calib $R/tools/ubench/hbm_calib.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/cal_$c -o pmc -- /tmp/hbm_calib > /tmp/cal_$c.log 2>&1
done
python3 - <<'PY' > $O/hbm_calib.txt
import sqlite3
GiB = float(1 << 30)
known = {"rd4": ("FETCH_SIZE", GiB), "rd16": ("FETCH_SIZE", GiB), "wr4": ("WRITE_SIZE", GiB), "wr16": ("WRITE_SIZE", GiB), "wr4s": ("WRITE_SIZE", (1 << 30) // 52 * 4.0)}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect("/tmp/cal_%s/pmc_results.db" % c)
    for name, n, avg in db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (c,)):
        k = name.split("(")[0]
        line = "%-10s %-6s dispatches %d  avg %.1f KiB = %.1f MiB" % (c, k, n, avg, avg / 1024.0)
        if k in known and known[k][0] == c:
            line += "   known %.1f MiB -> counter / known = %.3f" % (known[k][1] / 1048576.0, avg * 1024.0 / known[k][1])
        print(line)
PY
cat $O/hbm_calib.txt
