#!/usr/bin/env python3
"""tests/plan_sim/plan_sim.py -- run the product's gather planner on the CPU (no GPU needed) and print what it
would stage: tiles per shape, fetched bytes per plane (vs the source plane = over-fetch before any L2 reuse),
LDS bytes, pitch histogram and the modelled ds_read_b64 bank-conflict cycles.

    python tests/plan_sim/plan_sim.py [--config 2] [--plane 0|1] [--pieces 12] [--wide 200] [--strip 0] ...
The LUT comes from the CPU oracle's map (test infrastructure) quantised the way cv::remap does.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def build():
    so = os.path.join(HERE, "libplansim.so")
    srcs = [os.path.join(HERE, "plan_sim.cpp"), os.path.join(ROOT, "transform360_amd", "csrc", "t360_plan.cpp"),
            os.path.join(ROOT, "transform360_amd", "csrc", "t360_filtercfg.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs + [
            os.path.join(ROOT, "transform360_amd", "csrc", "t360_plan.h"),
            os.path.join(ROOT, "transform360_amd", "csrc", "t360_internal.h"),
            os.path.join(ROOT, "transform360_amd", "csrc", "t360_filtercfg.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "transform360_amd", "csrc")] + srcs + ["-o", so])
    L = C.CDLL(so)
    L.t360_plan_sim.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]
    return L


def lut_for(config, plane):
    import bench
    from oracle import t360_oracle as O
    from transform360_amd import handler
    from transform360_amd.abi import config_output, filter_defaults
    wl = bench.workload(config)
    ctx = filter_defaults(**wl["ov"])
    out_w, out_h = config_output(wl["in_w"], wl["in_h"], wl["edge"], ctx.output_layout, ctx.input_stereo_format,
                                 ctx.output_stereo_format)
    lin, lout = handler.FrameLayout(wl["in_w"], wl["in_h"]), handler.FrameLayout(out_w, out_h)
    o = O.Oracle(ctx, threads=8)
    k = 1 if plane else 0
    assert o.generateMapForPlane(*lin.dims[k], *lout.dims[k], k)
    m = o.map(k)
    q, nn = O.quantize_map(m)
    interp = int(ctx.interpolation_alg)
    ks = {0: 1, 1: 2, 2: 4, 4: 8}[interp]
    lut = np.zeros(q.shape[:2] + (4,), np.int16)
    if ks == 1:
        lut[..., 0] = np.clip(nn[..., 0], -32768, 32767)
        lut[..., 1] = np.clip(nn[..., 1], -32768, 32767)
    else:
        lut[..., 0] = np.clip(q[..., 0], -32768, 32767)
        lut[..., 1] = np.clip(q[..., 1], -32768, 32767)
        lut[..., 2] = q[..., 2].astype(np.int16)
    return np.ascontiguousarray(lut), lin.dims[k], lout.dims[k], ks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--plane", type=int, default=0)
    ap.add_argument("--pieces", type=str, default="12")
    ap.add_argument("--wide", type=str, default="200")
    ap.add_argument("--strip", type=str, default="0")
    ap.add_argument("--row-pad", type=str, default="0", help="pad + 256 * row_align, e.g. 2048 = align 8")
    ap.add_argument("--bshift", type=str, default="0")
    a = ap.parse_args()
    L = build()
    lut, (sw, sh), (dw, dh), ks = lut_for(a.config, a.plane)
    print("config %d plane %d: %dx%d <- %dx%d, ks %d" % (a.config, a.plane, dw, dh, sw, sh, ks))
    for pieces in [int(v) for v in a.pieces.split(",")]:
        for wide in [int(v) for v in a.wide.split(",")]:
            for strip in [int(v) for v in a.strip.split(",")]:
                for pm in [int(v) for v in a.row_pad.split(",")]:
                  for bs in [int(v) for v in a.bshift.split(",")]:
                    st = (C.c_longlong * 80)()
                    ok = L.t360_plan_sim(lut.ctypes.data, dw, dh, sw, sh, ks, pieces, wide, strip, pm, bs, st)
                    assert ok
                    src = sw * sh
                    ntile = st[0] + st[1] + st[2] + st[3]
                    hist = {i: st[12 + i] for i in range(33) if st[12 + i]}
                    print("bshift %d pieces %2d wide %3d strip %3d pad %d: strips %d wide %d sq %d s16 %d direct %d (%d px) | fetched %.2f MB = %.3fx "
                          "src | LDS %.2f MB (%.3fx fetched) | max pieces %d | lines %.3fx src "
                          "| tables %.1f+%.1f MB" % (
                              bs, pieces, wide, strip, pm, st[0], st[1], st[2], st[3], st[4], st[7], st[5] / 1e6, st[5] / src,
                              st[6] / 1e6, st[6] / max(st[5], 1), st[11],
                              st[8] / src, st[9] / 1e6, st[10] / 1e6))
                    print("   tile sizes (pieces: tiles):", hist)


if __name__ == "__main__":
    main()
