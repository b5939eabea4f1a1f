#!/usr/bin/env python
"""The numbers DESIGN.md / README.md quote, generated from the committed bench logs of the final-HEAD GPU round (prose must
not drift ahead of the last measurement).  Reads profiles/<prefix>_bench*.log (+ the shape-scale lines), prints a markdown table, and with
--write replaces the text between `<!-- numbers:begin -->` and `<!-- numbers:end -->` in DESIGN.md and README.md.

    python tools/numbers_table.py [--prefix r05f] [--write]"""
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        i = line.find('{"metric"')
        if i >= 0:
            try:
                return json.loads(line[i:])
            except json.JSONDecodeError:
                continue
    return None


def fmt(d):
    r = d.get("roofline", {})
    return "%.2f M" % (d["value"] / 1e6), "%.4f" % d["ms_per_step"], ("%.4f" % r["kernel_ms"]) if r.get("kernel_ms") else "-"


def main(prefix, write):
    P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (prefix, name))  # noqa: E731
    rows = []
    d = last_json(P("bench.log"))
    rows.append(("**default command** `python bench.py` (8192 envs, contacts, PGS; 64 + 320 steps = 10 whole epochs)", *fmt(d), "`%s_bench.log`" % prefix))
    head = d
    d = last_json(P("bench_driver_cmd.log"))
    if d:
        rq = d.get("requested_region", {})
        rows.append(("driver's command `--steps 20 --warmup 5`: `value` = its whole-epoch block (the 20 requested steps alone - positions 0..19, the light start of an "
                     "epoch - run at %.2f M)" % (rq.get("value", 0) / 1e6), *fmt(d), "`%s_bench_driver_cmd.log`" % prefix))
    for line in open(P("bench_variants.log")):
        m = re.match(r"\[(.*?)\] (.*)", line)
        if not m:
            continue
        d = None
        i = m.group(2).find('{"metric"')
        if i >= 0:
            try:
                d = json.loads(m.group(2)[i:])
            except json.JSONDecodeError:
                d = None
        if d:
            rows.append(("`%s`" % m.group(1), *fmt(d), "`%s_bench_variants.log`" % prefix))
    for S in (64, 2048, 8192):
        p = os.path.join(ROOT, "profiles", "r05_shapes_bench_%d.log" % S)
        if os.path.exists(p):
            d = last_json(p)
            rows.append(("`--per-clip-shapes --num-shapes %d` (one body shape per clip, env i -> shape i %% %d)" % (S, S), *fmt(d), "`r05_shapes_bench_%d.log`" % S))
    ppo_log = os.path.join(ROOT, "profiles", "r05g_bench_ppo.log")  # (the rollout fusion landed after the r05f round: its own log)
    if os.path.exists(P("bench_ppo.log")) and not prefix.startswith("r05"):
        ppo_log = P("bench_ppo.log")
    d = last_json(ppo_log if os.path.exists(ppo_log) else P("bench_ppo.log"))
    ppo = ""
    if d:
        c = d["config"]
        ppo = ("\n\nFull PPO loop (`bench.py --ppo`, 8192 envs, `%s`): **fps step %.2f M** (rollout: T_play %.4f s per epoch), fps total %.3f M "
               "(T_update %.3f s per epoch: fp32 MLPs through rocBLAS, outside this path)." % (os.path.basename(ppo_log) if os.path.exists(ppo_log) else prefix + "_bench_ppo.log", c["fps_step"] / 1e6, c["T_play_s_per_epoch"], c["fps_total"] / 1e6, c["T_update_s_per_epoch"]))
    r = head["roofline"]
    VI = (r.get("valu") or {}).get("valu_issue", {})
    cb = head.get("cpu_baseline")
    out = ["| configuration (one MI355X) | env-steps/s | ms per step | physics kernel ms | log |", "|---|---|---|---|---|"]
    out += ["| %s | %s | %s | %s | %s |" % row for row in rows]
    txt = "\n".join(out)
    txt += ("\n\nRoofline of the default command (`roofline` of the line): physics kernel %.4f ms per launch; algorithmic %.2f MB per launch -> %.0f GB/s = **%.4f of the 8 TB/s HBM peak**; "
            "measured HBM traffic %.1f MB per launch (%.2f x algorithmic); fp32 vector: %.2f TFLOP/s = %.4f of 157.3; VALU issue while three waves are resident: one instruction per "
            "%.2f cycles per SIMD = %.2f of the guide's 2-cycle wave64 issue, %.2f of the %.2f cycles measured on this GPU at a measured clock (`profiles/r06_valu_issue.txt`); "
            "%.1f of 64 lanes active per VALU instruction%s."
            % (r["kernel_ms"], r["algorithmic_bytes_per_launch"] / 1e6, r["achieved"], r["frac"], (r.get("traffic") or 0) / 1e6, (r.get("traffic") or 0) / r["algorithmic_bytes_per_launch"],
               r.get("valu_tflops", 0), r.get("valu_frac", 0), VI.get("cycles_per_inst_per_simd_at_3_waves", 0), VI.get("frac_of_guide_ceiling", 0), VI.get("frac_of_measured_ceiling", 0),
               VI.get("ceiling_cycles_per_inst_measured", 0), r.get("valu", {}).get("lanes_active_per_valu_op", 0),
               ("; the kernel ran at %.0f MHz" % r["valu"]["kernel_clock_mhz"]) if r.get("valu", {}).get("kernel_clock_mhz") else ""))
    if cb:
        b = cb["by_num_envs"]
        txt += ("\n\n`cpu_baseline` (kind \"port\": the float64 C oracle + numpy task ops on the GPU box's host, %s): **%.1f k env-steps/s at 8192 envs**, %.1f k at 1024, %.1f k at 4; "
                "physics per thread %s env-steps/s at 4 / 1024 / 8192 envs (`scaling_ok` %s)."
                % (cb["host"], cb["value"] / 1e3, b["1024"]["value"] / 1e3, b["4"]["value"] / 1e3, " / ".join("%.0f" % b[k]["physics_env_steps_per_s_per_thread"] for k in ("4", "1024", "8192")), cb.get("scaling_ok")))
    txt += ppo
    print(txt)
    if write:
        for f in ("DESIGN.md", "README.md"):
            p = os.path.join(ROOT, f)
            s = open(p).read()
            a, b = s.find("<!-- numbers:begin -->"), s.find("<!-- numbers:end -->")
            if a >= 0 and b > a:
                s = s[:a] + "<!-- numbers:begin -->\n(generated by `python tools/numbers_table.py --prefix %s --write` from the committed logs)\n\n" % prefix + txt + "\n" + s[b:]
                open(p, "w").write(s)


if __name__ == "__main__":
    pre = sys.argv[sys.argv.index("--prefix") + 1] if "--prefix" in sys.argv else "r06f"
    main(pre, "--write" in sys.argv)
