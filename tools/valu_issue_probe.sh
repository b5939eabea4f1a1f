#!/bin/bash
# The VALU issue ceiling of a gfx950 SIMD in the SAME units bench.py quotes for the physics kernel (4 x SQ_WAVE_CYCLES / waves per SIMD / SQ_INSTS_VALU),
# from SQ counters of tools/ubench/valu_issue at controlled residency (one workgroup per CU, W waves on every SIMD): no clock assumption.
# usage: gpurun -- 'bash tools/valu_issue_probe.sh'  -> gpurun_out/valu_issue_counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$R/tools/ubench/valu_issue cu > $O/valu_issue_cu.txt 2>&1
rm -rf $O/pmc_ub
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_ub -o pmc -- $R/tools/ubench/valu_issue cu > $O/pmc_ub.log 2>&1
rm -rf $O/pmc_ub2
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc_ub2 -o pmc -- $R/tools/ubench/valu_issue cu > $O/pmc_ub2.log 2>&1
python - <<PY > $O/valu_issue_counters.txt 2>&1
import sqlite3, glob
rows = {}
for db in ("$O/pmc_ub/pmc_results.db", "$O/pmc_ub2/pmc_results.db"):
    c = sqlite3.connect(db)
    # dispatches in launch order: 3 warm-ups + 1 timed per (W, mode); keep every 4th
    q = c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection where kernel_name like '%kcu%' order by dispatch_id").fetchall()
    for did, kn, cn, v in q:
        rows.setdefault(did, {"kernel": kn})[cn] = rows.get(did, {}).get(cn, 0.0) + v
ids = sorted(rows)
print("dispatches:", len(ids))
k = 0
for w in (1, 2, 3, 4):
    for mode in ("8 independent v_fma_f32", "dependent v_fma_f32 chain"):
        d = rows[ids[4 * k + 3]]
        k += 1
        ms = None
        for l in open("$O/valu_issue_cu.txt"):
            if l.startswith("[cu] %-28s W = %d" % (mode, w)):
                ms = float(l.split("kernel ")[1].split(" ms")[0])
        if "GRBM_GUI_ACTIVE" in d and ms:
            print("[clock] %-28s W = %d: GRBM_GUI_ACTIVE %.4g summed over 8 XCDs / 8 / %.3f ms (HIP events of the un-profiled run) = %.0f MHz; %d x %d instructions per SIMD in %.3f ms = %.3f ns = %.2f cycles at that clock"
                  % (mode, w, d["GRBM_GUI_ACTIVE"], ms, d["GRBM_GUI_ACTIVE"] / 8 / ms / 1e3, w, 8 * 32768, ms, ms * 1e6 / (w * 8 * 32768), ms * 1e6 / (w * 8 * 32768) * d["GRBM_GUI_ACTIVE"] / 8 / ms / 1e6))
        if "SQ_INSTS_VALU" in d and "SQ_WAVE_CYCLES" in d:
            print("[counters] %-28s W = %d: SQ_WAVES %.0f SQ_INSTS_VALU %.4g SQ_WAVE_CYCLES %.4g SQ_BUSY_CYCLES %.4g -> 4 x SQ_WAVE_CYCLES / W / SQ_INSTS_VALU = %.3f cycles per VALU instruction per SIMD%s"
                  % (mode, w, d.get("SQ_WAVES", 0), d["SQ_INSTS_VALU"], d["SQ_WAVE_CYCLES"], d.get("SQ_BUSY_CYCLES", 0), 4.0 * d["SQ_WAVE_CYCLES"] / w / d["SQ_INSTS_VALU"],
                     (" | GRBM_GUI_ACTIVE %.4g, SQ_ACTIVE_INST_VALU %.4g" % (d.get("GRBM_GUI_ACTIVE", 0), d.get("SQ_ACTIVE_INST_VALU", 0))) if "GRBM_GUI_ACTIVE" in d else ""))
PY
cat $O/valu_issue_cu.txt $O/valu_issue_counters.txt
rm -rf $O/pmc_ub $O/pmc_ub2
