#!/usr/bin/env python
"""Timeline of the (substep, env pair) jobs of one launch of the production physics kernel, from a V2P_LL_TIMELINE build run with
V2P_DEBUG=1 V2P_WAVE_TIMES=<file> (tools/mkvariant.sh timeline with V2P_FLAGS_PHYSICS_LL="-O3 -DV2P_LL_TIMELINE")."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0  # microseconds
bid, sjob, mono, ksum = a[:, 2] & 0xffffff, (a[:, 2] >> 24) & 15, (a[:, 2] >> 28) & 1, a[:, 2] >> 32
dur = end - start
cyc = a[:, 3].astype(float)
mhz = cyc / np.maximum(dur, 1e-3)
print("shader clock seen by the jobs (cycles / wall time): mean %.0f MHz, p5 %.0f, p95 %.0f; jobs that start in the first 100 us %.0f MHz, after 450 us %.0f MHz"
      % (mhz.mean(), *np.percentile(mhz, [5, 95]), mhz[start < 100].mean(), mhz[start > 450].mean() if (start > 450).any() else 0))
print("cycles per job: mono mean %.0f k max %.0f k; cut mean %.0f k max %.0f k" % (cyc[mono == 1].mean() / 1e3, cyc[mono == 1].max() / 1e3, cyc[mono == 0].mean() / 1e3, cyc[mono == 0].max() / 1e3))
print("jobs %d (mono %d, cut %d)  makespan %.1f us  sum(dur) %.0f us = %.1f us per slot at 3072 slots" % (len(a), mono.sum(), (1 - mono).sum(), end.max(), dur.sum(), dur.sum() / 3072))
m = mono == 1
print("mono jobs: dur mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; start max %.1f; end p50 %.1f p99 %.1f max %.1f"
      % (dur[m].mean(), *np.percentile(dur[m], [50, 90, 99]), dur[m].max(), start[m].max(), *np.percentile(end[m], [50, 99]), end[m].max()))
for s in range(int(sjob.max()) + 1):
    c = (~m) & (sjob == s)
    if c.any():
        print("cut jobs of substep %d: n %d  dur mean %.1f p99 %.1f max %.1f  start p1 %.1f p50 %.1f p99 %.1f  end p50 %.1f p99 %.1f max %.1f"
              % (s, c.sum(), dur[c].mean(), np.percentile(dur[c], 99), dur[c].max(), *np.percentile(start[c], [1, 50, 99]), *np.percentile(end[c], [50, 99]), end[c].max()))
ts = np.linspace(0, end.max(), 25)
print("running jobs at t (us):", [(int(t), int(((start <= t) & (end > t)).sum())) for t in ts])
late = np.argsort(end)[-10:]
print("last finishers (pair, substep, mono, start, dur):", [(int(bid[i]), int(sjob[i]), int(mono[i]), round(float(start[i]), 1), round(float(dur[i]), 1)) for i in late])
# per pair: when does its last job end, and how long did its jobs wait in between
cut_pairs = np.unique(bid[~m])
ends = np.array([end[(bid == p) & ~m].max() for p in cut_pairs[:4000]])
print("cut pairs: last job ends p50 %.1f p90 %.1f p99 %.1f max %.1f" % (*np.percentile(ends, [50, 90, 99]), ends.max()))
