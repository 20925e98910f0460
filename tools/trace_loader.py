#!/usr/bin/env python3
"""T360_TRACE dump taken with T360_DEBUG=16: where the loader wave spends the frame loop."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t = t[t[:, 7] > 0]
tot = (t[:, 7] - t[:, 0]) / 100.0
w, b, i = t[:, 1] / 100.0, t[:, 2] / 100.0, t[:, 3] / 100.0
nj, K = t[:, 4] // 1000, t[:, 4] % 1000
cb, cw = t[:, 5] / 100.0, t[:, 6] / 100.0
print("consumer wave 0: barrier-wait %.2f us, gather+store %.2f us (per workgroup)" % (cb.mean(), cw.mean()))
print("workgroups", len(t), "mean total %.2f us; loader: vmcnt-wait %.2f  barrier-wait %.2f  dma-issue %.2f" % (tot.mean(), w.mean(), b.mean(), i.mean()))
for k in sorted(set(K)):
    m = K == k
    print("  K=%d: %5d wgs, nj mean %.1f, total %.2f, wait %.2f, barrier %.2f, issue %.2f" % (k, m.sum(), nj[m].mean(), tot[m].mean(), w[m].mean(), b[m].mean(), i[m].mean()))
