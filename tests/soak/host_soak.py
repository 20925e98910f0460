#!/usr/bin/env python3
"""tests/soak/host_soak.py [iterations] -- host-pointer calls on buffers that come and go: arrays are allocated, used two or
three times (the library pins a buffer on its second sighting), freed, and new ones of other sizes take their addresses.
Every output is compared with the oracle: a stale pinned range would show up as wrong pixels (development soak)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import t360_oracle as O  # noqa: E402
from transform360_amd import handler as T  # noqa: E402
from transform360_amd.abi import filter_defaults  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(5)
ctx = filter_defaults(enable_low_pass_filter=0)
shapes = [((960, 480), (384, 256)), ((1280, 640), (768, 512)), ((640, 320), (192, 128))]
oracles, handles = {}, {}
for (iw, ih), (ow, oh) in shapes:
    o = O.Oracle(ctx, threads=4)
    assert o.generateMapForPlane(iw, ih, ow, oh, 0)
    t = T.VideoFrameTransform(ctx)
    assert t.generateMapForPlane(iw, ih, ow, oh, 0)
    oracles[(iw, ih)] = o
    handles[(iw, ih)] = t
bad = 0
live = []
for it in range(iters):
    (iw, ih), (ow, oh) = shapes[int(rng.integers(len(shapes)))]
    pad = int(rng.choice([0, 0, 32, 96]))
    if live and rng.random() < 0.5:
        src_full, dst_full, key, pad2, odims = live[int(rng.integers(len(live)))]
        if key != (iw, ih):
            (iw, ih), (ow, oh), pad = key, odims, pad2
        else:
            pad = pad2
    else:
        src_full = np.empty((ih, iw + pad), np.uint8)
        dst_full = np.empty((oh, ow + pad), np.uint8)
        live.append((src_full, dst_full, (iw, ih), pad, (ow, oh)))
        if len(live) > 6:
            live.pop(int(rng.integers(len(live))))  # freed: its pages go back to the allocator
    src_full[:] = rng.integers(0, 256, src_full.shape, dtype=np.uint8)
    dst_full[:] = 0x5A
    src, dst = src_full[:, :iw], dst_full[:, :ow]
    want = np.zeros((oh, ow), np.uint8)
    assert oracles[(iw, ih)].transformFramePlane(src, want, 0)
    assert handles[(iw, ih)].transformFramePlane(src, dst, 0, 0)
    if not np.array_equal(dst, want) or (pad and not (dst_full[:, ow:] == 0x5A).all()):
        bad += 1
        print("iteration %d: %dx%d pad %d differs in %d px" % (it, iw, ih, pad, int(np.count_nonzero(dst != want))), flush=True)
print("%d host-pointer calls, %d wrong" % (iters, bad))
sys.exit(1 if bad else 0)
