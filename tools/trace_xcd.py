#!/usr/bin/env python3
"""T360_TRACE dump taken with T360_DEBUG=32: per-XCD / per-CU load balance of the tiled kernel."""
import sys
from collections import defaultdict
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
t = t[t[:, 7] > 0]
t0 = int(t[:, 0].min())
xcd = (t[:, 6] >> np.uint64(32)).astype(int)
cu = (t[:, 6] & np.uint64(0xff00)).astype(int)
start = (t[:, 0].astype(np.int64) - t0) / 100.0
end = (t[:, 7].astype(np.int64) - t0) / 100.0
print("kernel span %.1f us, %d workgroups" % (end.max(), len(t)))
for x in sorted(set(xcd)):
    m = xcd == x
    print("xcd %d: %5d wgs  first start %6.1f  last start %6.1f  last end %6.1f  busy(sum of lifetimes) %8.0f us  mean life %.1f" %
          (x, m.sum(), start[m].min(), start[m].max(), end[m].max(), (end[m] - start[m]).sum(), (end[m] - start[m]).mean()))
# per-CU finish spread
fin = defaultdict(float)
for x, c, e in zip(xcd, cu, end):
    fin[(x, c)] = max(fin[(x, c)], e)
f = np.array(list(fin.values()))
print("per-CU last end: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (f.min(), np.percentile(f, 10), np.percentile(f, 50), np.percentile(f, 90), f.max()))
# gap between a WG's end and the next WG start on the same CU slot is not observable directly; report
# time-averaged residency in windows
for lo in range(0, int(end.max()), 50):
    hi = lo + 50
    ov = np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None).sum() / 50.0
    print("  window %3d-%3d us: mean resident workgroups %.0f (%.2f per CU)" % (lo, hi, ov, ov / 256))
