#!/usr/bin/env python
"""Stamp a profile summary (valu_summary.json / pmc_summary.json from the GPU box, where .git does not travel) with the git HEAD of the
tree it was collected from and copy it to its place under profiles/.  Refuses when the kernel sources of the working tree no longer hash to
the file's kernel_source_sha16 (the counters would describe another kernel).   usage: tools/stamp_profile.py <in.json> <out.json> [source note]"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from vid2player3d_amd import build  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
j = json.load(open(src))
now = build.kernel_source_hash()
if j.get("kernel_source_sha16") != now:
    raise SystemExit("%s: kernel_source_sha16 %s != the working tree's %s" % (src, j.get("kernel_source_sha16"), now))
j["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"]).decode().strip()
j["git_dirty"] = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "vid2player3d_amd/csrc"]).decode().strip())
if len(sys.argv) > 3:
    j["source"] = sys.argv[3]
json.dump(j, open(dst, "w"), indent=1)
print("%s -> %s (kernel sources %s, HEAD %s%s)" % (src, dst, now, j["git_head"], ", csrc dirty" if j["git_dirty"] else ""))
