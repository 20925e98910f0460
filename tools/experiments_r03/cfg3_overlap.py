#!/usr/bin/env python3
"""Does config 3 gain from overlapping the low-pass of one half of a batch with the gather of the other?
H handles (own streams), each 64/H frames of the same step; wall clock over K steps.  Development experiment."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from transform360_amd import handler  # noqa: E402
from transform360_amd.abi import config_output, filter_defaults  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    wl = bench.workload(cfg)
    ctx = filter_defaults(**wl["ov"])
    ow, oh = config_output(wl["in_w"], wl["in_h"], wl["edge"], ctx.output_layout, ctx.input_stereo_format, ctx.output_stereo_format)
    lin, lout = handler.FrameLayout(wl["in_w"], wl["in_h"]), handler.FrameLayout(ow, oh)
    F = 64
    d_in = torch.empty(F * lin.frame_bytes, dtype=torch.uint8, device="cuda")
    for j in range(F):
        handler.fill_noise(d_in[j * lin.frame_bytes:(j + 1) * lin.frame_bytes], handler.frame_seed(j))
    ref = None
    for H in (1, 2, 4):
        n = F // H
        hs = []
        for _ in range(H):
            t = handler.VideoFrameTransform(ctx)
            for idx in (0, 1):
                assert t.generateMapForPlane(*lin.dims[idx], *lout.dims[idx], idx)
            hs.append(t)
        descs = hs[0].plane_descs(lin, lout)
        d_out = torch.zeros(F * lout.frame_bytes, dtype=torch.uint8, device="cuda")

        def step():
            for k, t in enumerate(hs):
                assert t.transformFrames(d_in[k * n * lin.frame_bytes:], lin.frame_bytes, d_out[k * n * lout.frame_bytes:], lout.frame_bytes, n, descs)

        for _ in range(2000):
            step()
        for t in hs:
            t.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200):
                step()
            for t in hs:
                t.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
        s = int(d_out.to(torch.int64).sum())
        ref = s if ref is None else ref
        print("cfg %d: %d handle(s) x %d frames: %.4f ms per 64 frames, kernel %s, checksum %s" % (cfg, H, n, best, hs[0].lastKernel(), "same" if s == ref else "DIFFERENT"), flush=True)
        for t in hs:
            t.close()


main()
