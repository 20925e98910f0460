#!/usr/bin/env python3
"""tests/plan_sim/request_model.py -- TCP -> TCC read requests of the gather's staging, from the plan alone (VERDICT round 5,
item 1a): per tile class the 1 KiB DMA instructions, the distinct 128-byte lines under each instruction's 64 chunks (what the
CU's L1 asks the L2 for) per staged KiB, and the distinct lines per tile; summed over the planes of a frame and multiplied by
the frames of a launch, next to the measured TCP_TCC_READ_REQ of the profiled launch.

    python tests/plan_sim/request_model.py [--config 2] [--frames 64] [--measured 14176454]"""
import argparse
import ctypes as C

import plan_sim

KINDS = {0: "32x32", 1: "16x16", 3: "128x8", 4: "64x16", 5: "128x16", 6: "256x8", 7: "scatter", 8: "all staged"}


def model(L, config, plane, pieces, waves, cost_lines=0):
    lut, (sw, sh), (dw, dh), ks = plan_sim.lut_for(config, plane)
    out = (C.c_longlong * 36)()
    L.t360_plan_requests.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]
    n = L.t360_plan_requests(lut.ctypes.data, dw, dh, sw, sh, ks, pieces, waves, cost_lines, out)
    assert n >= 0
    return {k: tuple(out[k * 4:k * 4 + 4]) for k in KINDS}, sw * sh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--pieces", type=int, default=24)
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--measured", type=float, default=14176454.1, help="TCP_TCC_READ_REQ of the profiled launch (profiles/r05_pmc_summary.txt)")
    a = ap.parse_args()
    L = plan_sim.build()
    total_req = total_lines = total_pieces = 0
    for plane, copies in ((0, 1), (1, 2)):
        rows, src = model(L, a.config, plane, a.pieces, a.waves, 1 if a.config == 1 else 0)
        print("plane %d (x%d per frame), source %d bytes:" % (plane, copies, src))
        for k, (tiles, pieces, req, lines) in rows.items():
            if tiles:
                print("  %-10s %5d tiles %6d KiB staged  %7d requests = %5.2f per staged KiB (whole lines would be 8.00)  %7d distinct lines per tile = %5.2f per KiB"
                      % (KINDS[k], tiles, pieces, req, req / pieces, lines, lines / pieces))
        total_req += copies * rows[8][2]
        total_lines += copies * rows[8][3]
        total_pieces += copies * rows[8][1]
    print("frame: %d KiB staged, %d requests by instruction (%.2f per KiB = %.3f GB per %d frames at 128 B), %d by tile" % (
        total_pieces, total_req, total_req / total_pieces, total_req * a.frames * 128 / 1e9, a.frames, total_lines))
    return_line = None
    print("launch of %d frames: %.2f M requests modelled (by instruction) / %.2f M (by tile) vs %.2f M measured (TCP_TCC_READ_REQ): model / measured = %.3f / %.3f" % (
        a.frames, total_req * a.frames / 1e6, total_lines * a.frames / 1e6, a.measured / 1e6, total_req * a.frames / a.measured, total_lines * a.frames / a.measured))


def variants():
    """the shipped plan against the variants in the tree (256x8 tiles, line-cost shapes, scatter tiles, the 4-wave plan):
    requests per frame by instruction; nothing below a 10 % cut gets GPU time"""
    L = plan_sim.build()
    base = None
    for name, pieces, waves in (("shipped: 8 waves, 24 pieces", 24, 8), ("256x8 tiles wherever they fit", 24, 8 | (1000 << 8)),
                                ("shapes compared by lines", 24, 8 | (1 << 20)), ("scatter tiles, strips of 4 lines", 24, 8 | (4 << 24)),
                                ("scatter tiles, strips of 2 lines", 24, 8 | (2 << 24)), ("4 waves, 12 pieces", 12, 4)):
        req = pieces_ = 0
        for plane, copies in ((0, 1), (1, 2)):
            rows, _ = model(L, 2, plane, pieces, waves)
            req += copies * rows[8][2]
            pieces_ += copies * rows[8][1]
        base = base or req
        print("%-36s %7d requests per frame (%+5.1f %%), %5.2f per staged KiB, %6d KiB staged" % (name, req, 100.0 * (req - base) / base, req / pieces_, pieces_))


if __name__ == "__main__":
    import sys
    if "--variants" in sys.argv:
        variants()
    else:
        main()
