#!/usr/bin/env python
"""Analyse the per-wave wall-clock stamps written by V2P_DEBUG=1 V2P_WAVE_TIMES=<file> (physics_ll_kernel, last launch of the run)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
start, end, packed = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2]  # microseconds (100 MHz wall clock)
key, tsum, tmax, kpred = packed % 1024, (packed // 1024) % 1024, (packed // 1048576) % 1024, packed // 1073741824
dur = end - start
print("waves %d  makespan %.1f us  sum(dur)/2048 slots %.1f us  mean dur %.1f  max dur %.1f  p50 %.1f p90 %.1f p99 %.1f"
      % (len(a), end.max(), dur.sum() / 2048, dur.mean(), dur.max(), *np.percentile(dur, [50, 90, 99])))
order = np.arange(len(a))
print("launch-order deciles: mean start / mean dur / mean key(touched links)")
for q in range(10):
    s = slice(q * len(a) // 10, (q + 1) * len(a) // 10)
    print("  %d: start %7.1f  dur %7.1f  end %7.1f  touched %.1f" % (q, start[s].mean(), dur[s].mean(), end[s].max(), (key[s] // 8).mean()))
late = np.argsort(end)[-8:]
print("last finishers (slot, start, dur, touched):", [(int(i), round(start[i], 1), round(dur[i], 1), int(key[i] // 8)) for i in late])
tk = key // 8
for k in sorted(set(tk.tolist())):
    m = tk == k
    print("  touched %2d: waves %5d  dur mean %.1f max %.1f" % (k, m.sum(), dur[m].mean(), dur[m].max()))
# concurrency over time
ts = np.linspace(0, end.max(), 21)
print("running waves at t:", [(round(t, 0), int(((start <= t) & (end > t)).sum())) for t in ts])

# how well do candidate keys predict the wave's duration (half 0's env only)?
def r2(x):
    x = x.astype(float)
    A = np.stack([x, np.ones_like(x)], 1)
    coef, res, *_ = np.linalg.lstsq(A, dur, rcond=None)
    return 1 - ((A @ coef - dur) ** 2).sum() / ((dur - dur.mean()) ** 2).sum(), coef
first = start < 50  # first-round waves ran with the same neighbours all along
for name, x in (("predicted key (prev launch, touched*8+depth)", kpred), ("touched links, last substep", key // 8), ("sum of touched links over the substeps", tsum),
                ("max touched links over the substeps", tmax)):
    print("R2 of duration vs %-48s all %.3f  first round %.3f" % (name, r2(x)[0], 1 - 0 if first.sum() < 3 else (lambda xx, dd: 1 - ((np.polyval(np.polyfit(xx, dd, 1), xx) - dd) ** 2).sum() / ((dd - dd.mean()) ** 2).sum())(x[first].astype(float), dur[first])))
print("fit dur = a*sum_touched + b:", r2(tsum)[1])
