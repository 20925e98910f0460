#!/usr/bin/env python3
"""tools/trace_xcd_balance.py FILE -- from a T360_TRACE dump of the instrumented library: when does each XCD finish,
and how long do workgroups of each tile size live (us per frame vs pieces)?"""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
T = (a[:, :6].astype(np.float64) - float(t0)) / 100.0  # us
kind = (a[:, 6] >> np.uint64(32)).astype(int)
pieces = (a[:, 6] & np.uint64(0xffff)).astype(int)
xcc = (a[:, 7] >> np.uint64(32)).astype(int)
m = T[:, 5] > 0
print("kernel span %.1f us, %d staged workgroups" % (T[m, 5].max(), m.sum()))
for x in sorted(set(xcc[m])):
    mx = m & (xcc == x)
    life = T[mx, 5] - T[mx, 0]
    print("xcd %d: workgroups %4d  last end %.1f us  sum of lives %.0f us  pieces %d" % (x, mx.sum(), T[mx, 5].max(), life.sum(), pieces[mx].sum()))
life = T[m, 5] - T[m, 0]
long_ = m.copy()
print("life (us) of the 64-frame workgroups by pieces (median, n):")
big = life > 0.5 * np.percentile(life, 90)
for p in sorted(set(pieces[m])):
    sel = (pieces[m] == p) & big
    if sel.sum() >= 4:
        print("  pieces %2d: %.1f us  n=%d" % (p, np.median(life[sel]), sel.sum()))
