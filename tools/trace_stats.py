#!/usr/bin/env python3
"""Summarise a T360_TRACE dump: per-workgroup phase durations in microseconds (100 MHz clock)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
ok = (t > 0).all(axis=1)
t = t[ok]
t0 = t[:, 0].min()
us = lambda x: x / 100.0
print("workgroups", len(t), "kernel span %.1f us" % us(t[:, 7].max() - t0))
names = [("start->loader setup done", 0, 1), ("loader: prologue DMA issue", 1, 2), ("loader: issue->frame0 landed", 2, 3),
         ("start->LUT+weights in regs", 0, 4), ("pixels ready->first frame done", 4, 5), ("first frame done->end (consumer)", 5, 7),
         ("whole workgroup", 0, 7)]
for n, a, b in names:
    d = us(t[:, b] - t[:, a])
    print("%-36s mean %8.2f  p10 %8.2f  p50 %8.2f  p90 %8.2f  max %8.2f" % (n, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
start = us(t[:, 0] - t0)
print("start times: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(start, [10, 50, 90, 100])))
