#!/usr/bin/env python3
"""tools/trace_stats.py FILE -- summarise the per-workgroup timestamps the INSTRUMENTED library writes with
T360_TRACE=FILE (8 x u64 per workgroup, 100 MHz wall clock): phase durations and resident workgroups per CU."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
T = (a[:, :6].astype(np.float64) - float(t0)) / 100.0  # us
kind = (a[:, 6] >> np.uint64(32)).astype(int)
pieces = (a[:, 6] & np.uint64(0xffff)).astype(int)
xcc = (a[:, 7] >> np.uint64(32)).astype(int)
hw = (a[:, 7] & np.uint64(0xffffffff)).astype(int)
cu = xcc * 65536 + (hw & 0xff00)
print("workgroups traced: %d, kernel span %.1f us" % (len(a), T[:, 5].max()))
for k, name in ((5, "128x16"), (6, "256x8"), (4, "64x16"), (0, "32x32"), (3, "128x8"), (1, "16x16")):
    m = (kind == k) & (T[:, 5] > 0)
    if not m.any():
        continue
    d = T[m]
    print("%-6s n=%5d  desc->chunks %.2f  ->setup %.2f  ->frame0 %.2f  frame1 %.2f  rest %.2f  life %.2f us (medians); pieces %.1f" % (
        name, m.sum(), np.median(d[:, 1] - d[:, 0]), np.median(d[:, 2] - d[:, 1]), np.median(d[:, 3] - d[:, 2]),
        np.median(d[:, 4] - d[:, 3]), np.median(d[:, 5] - d[:, 4]), np.median(d[:, 5] - d[:, 0]), pieces[m].mean()))
m = T[:, 5] > 0
# resident workgroups per CU over time (sampled)
span = T[m, 5].max()
samples = np.linspace(0.05 * span, 0.9 * span, 40)
res = []
for s in samples:
    live = m & (T[:, 0] <= s) & (T[:, 5] > s)
    res.append(live.sum() / max(1, len(np.unique(cu[m]))))
print("resident workgroups per CU (mean over time): %.2f on %d CUs" % (np.mean(res), len(np.unique(cu[m]))))
# gap between a workgroup's end and the next start on the same CU slot is not observable directly; report start rate
starts = np.sort(T[m, 0])
print("workgroup starts: first %.1f us: %d, steady rate %.1f per us" % (5.0, (starts < 5).sum(), len(starts) / span))
# machine-wide phase occupancy over time: how many workgroups are in their prologue (before frame 0) vs streaming
edges = np.linspace(0, span, 60)
pro, run = [], []
for s in edges:
    live = m & (T[:, 0] <= s) & (T[:, 5] > s)
    inpro = live & (T[:, 3] > s)
    pro.append(int(inpro.sum()))
    run.append(int((live & ~inpro).sum()))
print("time(us): workgroups in prologue / streaming")
print(" ".join("%d:%d/%d" % (int(e), p, r) for e, p, r in zip(edges, pro, run)))
