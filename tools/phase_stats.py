#!/usr/bin/env python3
"""tools/phase_stats.py FILE -- T360_PHASES dump of the instrumented library: where the waves of the tiled gather spend
their cycles per frame (wave 0 and the last wave of every workgroup)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2, 8).astype(np.float64)
names = ["dma wait", "barrier", "store+issue", "gather", "rest"]
for w, label in ((0, "wave 0"), (1, "last wave")):
    m = a[:, w, 5] >= 32
    d = a[m, w]
    nf = d[:, 5]
    per = d[:, :5] / nf[:, None]
    tot = per.sum(axis=1)
    print("%s: %d workgroups of >= 32 frames; cycles per frame: total median %.0f (p10 %.0f, p90 %.0f)" % (
        label, m.sum(), np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
    for k, n in enumerate(names):
        print("   %-12s median %6.0f  mean %6.0f  (%.0f%%)" % (n, np.median(per[:, k]), per[:, k].mean(), 100 * per[:, k].mean() / tot.mean()))
    pieces = (a[m, w, 6].astype(np.uint64) & np.uint64(0xffff)).astype(int)
    for lo, hi in ((1, 7), (8, 10), (11, 14), (15, 32)):
        s = (pieces >= lo) & (pieces <= hi)
        if s.sum():
            print("   pieces %2d-%2d (n=%4d): " % (lo, hi, s.sum()) + "  ".join("%s %.0f" % (n.split()[0], per[s, k].mean()) for k, n in enumerate(names)))
