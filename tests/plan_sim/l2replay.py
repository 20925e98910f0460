#!/usr/bin/env python3
"""tests/plan_sim/l2replay.py TRACE.bin -- replay a T360_TRACE dump of the instrumented library (64-frame cfg2 launch)
through the L2 model of l2sim.cpp with the MEASURED start/end time of every workgroup."""
import ctypes as C
import sys

import numpy as np

import l2sim
import plan_sim

path = sys.argv[1]
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
fpb, tail_pct, tail_frames = 64, 12, 16
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
L = l2sim.build()
L.t360_l2replay.restype = C.c_longlong
ly, (swy, shy), (dwy, dhy), ks = plan_sim.lut_for(2, 0)
lc, (swc, shc), (dwc, dhc), _ = plan_sim.lut_for(2, 1)
st = (C.c_longlong * 16)()
# first call with no items: tile and direct counts
z = (C.c_int * 1)()
zd = (C.c_double * 1)()
L.t360_l2replay(C.c_void_p(ly.ctypes.data), dwy, dhy, swy, shy, C.c_void_p(lc.ctypes.data), dwc, dhc, swc, shc, ks, 24, 8, 2, 4 << 20, 16,
                0, z, z, z, z, zd, zd, st)
total_tiles, total_direct = st[2], st[3]
groups = (nframes + fpb - 1) // fpb
tail_frames = max(1, min(fpb, tail_frames))
tail_groups = (nframes + tail_frames - 1) // tail_frames
tail_percent = tail_pct if tail_groups > groups else 0
direct_blocks = (total_direct * groups + 7) & ~7
print("tiles %d direct %d direct_blocks %d trace entries %d" % (total_tiles, total_direct, direct_blocks, len(a)))
items = []
t_min = a[a[:, 0] > 0][:, 0].min()
for wg in range(direct_blocks, len(a)):
    if a[wg, 5] == 0:
        continue
    i = wg - direct_blocks
    xcd, k = i & 7, i >> 3
    q, rem = total_tiles >> 3, total_tiles & 7
    ln = q + (1 if xcd < rem else 0)
    start = xcd * q + min(xcd, rem)
    len_tail = (ln * tail_percent) // 100
    len_head = ln - len_tail
    if k < len_head * groups:
        tl, g = divmod(k, groups)
        b, f = start + tl, fpb
    else:
        k2 = k - len_head * groups
        tl, g = divmod(k2, tail_groups)
        if tl >= len_tail:
            continue
        b, f = start + len_head + tl, tail_frames
    f0 = g * f
    f1 = min(f0 + f, nframes)
    xcc = int(a[wg, 7] >> np.uint64(32))
    # requests of frame f0 go out at mark 1..2 (prologue DMA), the last frame's about two frames before the end
    t0 = (float(a[wg, 2]) - float(t_min)) / 100.0
    t1 = (float(a[wg, 5]) - float(t_min)) / 100.0
    items.append((b, f0, f1, xcc, t0, t1))
items = np.array(items, dtype=np.float64)
print("items %d, span %.1f us; xcc ids seen %s" % (len(items), items[:, 5].max(), sorted(set(items[:, 3].astype(int)))))
n = len(items)
arr = lambda col, ct: (ct * n)(*[ct(v).value if ct is C.c_double else int(v) for v in items[:, col]])
for l2kb in (16384, 8192, 4096, 2048, 1024):
    r = L.t360_l2replay(C.c_void_p(ly.ctypes.data), dwy, dhy, swy, shy, C.c_void_p(lc.ctypes.data), dwc, dhc, swc, shc, ks, 24, 8, 2,
                        l2kb << 10, 16, n, arr(0, C.c_int), arr(1, C.c_int), arr(2, C.c_int), arr(3, C.c_int),
                        arr(4, C.c_double), arr(5, C.c_double), st)
    src = (swy * shy + 2 * swc * shc) * nframes
    print("L2 %d KiB: fabric reads %.2f M lines = %.3f GB = %.3fx src (hit %.1f%%)" % (l2kb, r / 1e6, r * 128 / 1e9, r * 128 / src,
                                                                                  100.0 * st[1] / (st[0] + st[1])))
