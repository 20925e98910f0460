#!/usr/bin/env python3
"""T360_TRACE dump taken with T360_DEBUG=32: co-resident workgroups per CU over time."""
import sys
from collections import defaultdict
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
t = t[t[:, 7] > 0]
ev = defaultdict(list)
for row in t:
    key = (int(row[6]) >> 32, int(row[6]) & 0xff00)
    ev[key].append((int(row[0]), 1))
    ev[key].append((int(row[7]), -1))
mx, area, span = [], 0.0, 0.0
for key, e in ev.items():
    e.sort()
    cur = m = 0
    last = e[0][0]
    for ts, d in e:
        area += cur * (ts - last)
        last = ts
        cur += d
        m = max(m, cur)
    span += e[-1][0] - e[0][0]
    mx.append(m)
print("CUs seen %d; co-resident workgroups per CU: max %d, mean of per-CU max %.2f, time-averaged %.2f" % (len(ev), max(mx), np.mean(mx), area / span))
