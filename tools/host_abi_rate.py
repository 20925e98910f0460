#!/usr/bin/env python3
"""PCIe-inclusive rate of the literal reference ABI (host pointers, one synchronous call per plane,
the ffmpeg filter's pattern vf_transform360.c:368-397) for BASELINE config 2.  Not the bench metric:
DESIGN.md quotes it next to the HBM-resident number."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transform360_amd import abi, handler  # noqa: E402

ctx = abi.filter_defaults(enable_low_pass_filter=0, interpolation_alg=abi.CUBIC)
in_w, in_h, out_w, out_h = 3840, 1920, 1536, 1024
rng = np.random.default_rng(1)
planes_in = [rng.integers(0, 256, (in_h, in_w), dtype=np.uint8)] + [
    rng.integers(0, 256, (in_h // 2, in_w // 2), dtype=np.uint8) for _ in range(2)]
planes_out = [np.zeros((out_h, out_w), np.uint8)] + [np.zeros((out_h // 2, out_w // 2), np.uint8) for _ in range(2)]
with handler.VideoFrameTransform(ctx) as t:
    assert t.generateMapForPlane(in_w, in_h, out_w, out_h, 0)
    assert t.generateMapForPlane(in_w // 2, in_h // 2, out_w // 2, out_h // 2, 1)
    def frame():
        for k in range(3):
            assert t.transformFramePlane(planes_in[k], planes_out[k], 1 if k else 0, k)
    for _ in range(3):
        frame()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        frame()
    dt = (time.perf_counter() - t0) / n
bytes_per_frame = sum(p.nbytes for p in planes_in) + sum(p.nbytes for p in planes_out)
print("host-pointer ABI, cfg2: %.3f ms/frame, %.0f frames/s, %.0f Mpix/s, %.2f GB/s over PCIe (in+out)" % (
    dt * 1e3, 1 / dt, out_w * out_h / dt / 1e6, bytes_per_frame / dt / 1e9))
