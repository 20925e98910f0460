#!/usr/bin/env python3
"""tests/plan_sim/l2sim.py -- offline model of the tiled gather's fabric reads (see l2sim.cpp).  Development tool."""
import argparse
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import plan_sim  # noqa: E402


def build():
    so = os.path.join(HERE, "libl2sim.so")
    srcs = [os.path.join(HERE, "l2sim.cpp"), os.path.join(ROOT, "transform360_amd", "csrc", "t360_plan.cpp")]
    deps = srcs + [os.path.join(ROOT, "transform360_amd", "csrc", h) for h in ("t360_plan.h", "t360_internal.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "transform360_amd", "csrc")] + srcs + ["-o", so])
    return C.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--pieces", type=int, default=24)
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--slots", type=str, default="64")
    ap.add_argument("--l2", type=str, default="4096", help="KiB")
    ap.add_argument("--ways", type=int, default=16)
    ap.add_argument("--jitter", type=str, default="0")
    ap.add_argument("--fpb", type=int, default=64, help="negative: frame-group major order")
    ap.add_argument("--lead", type=str, default="0", help="N: limit vs the slowest of the XCD; 100+N: vs resident neighbours")
    ap.add_argument("--window", type=str, default="8")
    ap.add_argument("--spread", type=int, default=0, help="+- percent of speed per work item")
    a = ap.parse_args()
    L = build()
    L.t360_l2sim.restype = C.c_longlong
    ly, (swy, shy), (dwy, dhy), ks = plan_sim.lut_for(a.config, 0)
    lc, (swc, shc), (dwc, dhc), _ = plan_sim.lut_for(a.config, 1)
    for slots in [int(v) for v in a.slots.split(",")]:
        for l2 in [int(v) for v in a.l2.split(",")]:
          for lead in [int(v) for v in a.lead.split(",")]:
           for win in [int(v) for v in a.window.split(",")]:
            for jit in [int(v) for v in a.jitter.split(",")]:
                st = (C.c_longlong * 16)()
                r = L.t360_l2sim(C.c_void_p(ly.ctypes.data), dwy, dhy, swy, shy, C.c_void_p(lc.ctypes.data), dwc, dhc, swc, shc, ks,
                                 a.pieces, a.waves, a.order, a.frames, slots, l2 * 1024, a.ways, jit, a.fpb, lead, a.spread, win, st)
                assert r >= 0
                src = st[4] * a.frames
                print("lead %d win %d waited %d us t %.1f us slots %3d l2 %5d KiB jitter %2d%%: tiles %d | staged %.3fx src | fabric reads %.1f M lines = %.3f GB = %.3fx src | hit %.1f%%" % (
                    lead, win, st[7], st[6] / 1000.0, slots, l2, jit, st[3], st[2] * 16 * a.frames / src, st[0] / 1e6, st[0] * 128 / 1e9, st[0] * 128 / src,
                    100.0 * st[1] / (st[0] + st[1])))


if __name__ == "__main__":
    main()
