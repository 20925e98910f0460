#!/usr/bin/env python3
"""tools/phase_stats.py FILE -- T360_PHASES dump of the instrumented library: where the waves of the tiled gather spend
their cycles per frame (wave 0 and the last wave of every workgroup)."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 32)
a = raw[:, :16].reshape(-1, 2, 8).astype(np.float64)
names = ["dma wait", "barrier", "store+issue", "gather", "rest"]
for w, label in ((0, "wave 0"), (1, "last wave")):
    m = a[:, w, 5] >= 8
    d = a[m, w]
    nf = d[:, 5]
    per = d[:, :5] / nf[:, None]
    tot = per.sum(axis=1)
    print("%s: %d workgroups of >= 8 frames; cycles per frame: total median %.0f (p10 %.0f, p90 %.0f)" % (
        label, m.sum(), np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
    for k, n in enumerate(names):
        print("   %-12s median %6.0f  mean %6.0f  (%.0f%%)" % (n, np.median(per[:, k]), per[:, k].mean(), 100 * per[:, k].mean() / tot.mean()))
    pieces = (a[m, w, 6].astype(np.uint64) & np.uint64(0xffff)).astype(int)
    for lo, hi in ((1, 7), (8, 10), (11, 14), (15, 32)):
        s = (pieces >= lo) & (pieces <= hi)
        if s.sum():
            print("   pieces %2d-%2d (n=%4d): " % (lo, hi, s.sum()) + "  ".join("%s %.0f" % (n.split()[0], per[s, k].mean()) for k, n in enumerate(names)))

# item timeline of wave 0: begin, ticket, set-up done, end (100 MHz ticks)
it = raw[:, 16:].reshape(-1, 4, 4).astype(np.float64)
ok = it[:, 0, 0] > 0
t0 = it[ok][:, 0, 0].min()
print("item timeline (us, medians over %d workgroups): kernel-relative begin | ticket | set-up | frames | gap to next item" % ok.sum())
for k in range(4):
    m = ok & (it[:, k, 3] > 0)
    if not m.sum():
        continue
    d = it[m, k]
    gap = (it[m, k + 1, 0] - d[:, 3]) / 100.0 if k < 3 else None
    gm = np.median(gap[it[m, k + 1, 0] > 0]) if k < 3 and (it[m, k + 1, 0] > 0).any() else float("nan")
    print("   item %d (n=%4d): begin %7.1f | ticket %5.2f | set-up %5.2f | frames %6.2f | gap %5.2f" % (
        k, m.sum(), np.median(d[:, 0] - t0) / 100.0, np.median(d[:, 1] - d[:, 0]) / 100.0, np.median(d[:, 2] - d[:, 1]) / 100.0,
        np.median(d[:, 3] - d[:, 2]) / 100.0, gm))
