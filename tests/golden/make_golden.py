#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/: maps and filter configurations FROM THE REFERENCE'S OWN CODE, frame
hashes from the reference's frame path over the oracle's cv:: restatement (SHIM-GENERATED, see below).

Run in the builder container (where /root/reference is mounted):

    python tests/golden/make_golden.py

It builds oracle/_ref (the reference's VideoFrameTransform.cpp compiled from /root/reference
against the test-only cv::Mat shim, see oracle/Makefile) and records, per named case of
tests/cases.py:

  maps.json     FNV-1a-64 of the reference's warp map (float bits / 1/32-px quanta / nearest
                picks, exactly the three hashes of SURVEY.md Appendix B) + sample points
  lowpass.json  the reference's low-pass segments: count, rectangles hash, kernel-bits hash,
                per-band integer (x256) kernels
  frames.json   FNV-1a-64 of output planes produced by the reference's frame path
                (transformFramePlane) -- its orchestration is the reference's own code; the cv::
                arithmetic underneath is the oracle restatement of OpenCV (parity UNPINNED at
                that boundary, see oracle/t360_oracle_cv.c)

The GPU box has no /root/reference: tests there compare against these committed files.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import t360_oracle as O  # noqa: E402
from tests import cases  # noqa: E402

# SURVEY.md Appendix B (produced by the surveyor's build of the reference): cross-checked below
SURVEY_APPENDIX_B = {
    "cfg1_luma": ("6797220f1c570869", "b1c1e6c06d122273", "374f3241e8921619"),
    "cfg1_chroma": ("3584fa90c3849f97", "e474a8c6d64ac5bf", "4c530a4beaf89d0c"),
    "cfg2_luma": ("94f4a0b1d826acbe", "c6896e5948236134", "2f8a95772caf9d95"),
    "cfg2_chroma": ("6797220f1c570869", "b1c1e6c06d122273", "374f3241e8921619"),
    "cfg4_luma": ("f9c55e5a8e337969", "c8a9af1195aad75c", "025b75b63a11e9b8"),
    "cfg4_chroma": ("ab77a481241fb7ab", "18a77ea4e85b60e7", "a230fc1f610c5cab"),
    "rotated": ("252be7f5ba840dee", "b8fbd51a75f64800", "5d86e78e539de0f8"),
}


def hx(v):
    return "%016x" % v


def map_record(m):
    q, nn = O.quantize_map(m)
    h, w = m.shape[:2]
    pts = [(0, 0), (h // 4, w // 6), (h // 2, w // 2), (h - 1, w - 1), (h // 2 + 7, w // 3 + 11)]
    return {
        "w": w, "h": h,
        "f32": hx(O.fnv1a64(m)), "q": hx(O.fnv1a64(q)), "nn": hx(O.fnv1a64(nn)),
        "samples": [[r, c, float(m[r, c, 0]).hex(), float(m[r, c, 1]).hex()] for r, c in pts],
    }


def lowpass_record(segs):
    rects = np.array([s[:4] for s in segs], np.int32).reshape(-1, 4)
    kbits = np.concatenate([np.concatenate([s[4], s[5]]) for s in segs]).astype(np.float32) if segs else np.zeros(0, np.float32)
    bands = {}
    for s in segs:
        if s[1] in bands:
            continue
        bands[s[1]] = {
            "top": s[1], "height": s[3], "tile_w": s[2],
            "kx_q8": [int(v) for v in np.rint(s[4].astype(np.float64) * 256)],
            "ky_q8": [int(v) for v in np.rint(s[5].astype(np.float64) * 256)],
            "kx_center_hex": float(s[4][len(s[4]) // 2]).hex(),
        }
    return {"count": len(segs), "rects": hx(O.fnv1a64(rects)), "kernels": hx(O.fnv1a64(kbits)),
            "bands": [bands[k] for k in sorted(bands)]}


def main():
    O.build(ref=True)
    assert O.ref_available(), "oracle/_ref could not be built: is /root/reference mounted?"

    maps = {}
    for name, (ov, dims) in {**cases.MAP_CASES, **cases.LAYOUT_MAP_CASES}.items():
        r = O.Ref(cases.make_ctx(ov))
        assert r.generateMapForPlane(*dims, 0)
        rec = map_record(r.map(0))
        if name in SURVEY_APPENDIX_B:
            assert (rec["f32"], rec["q"], rec["nn"]) == SURVEY_APPENDIX_B[name], name
            rec["survey_appendix_b"] = True
        maps[name] = rec
        r.close()

    lowpass = {}
    for name, (ov, dims) in cases.LOWPASS_CASES.items():
        r = O.Ref(cases.make_ctx(ov))
        assert r.generateMapForPlane(*dims, 0)
        lowpass[name] = lowpass_record(r.segments(0))
        r.close()

    frames = {}
    for name, (ov, dims, pin, pout) in {**cases.FRAME_CASES, **cases.LAYOUT_FRAME_CASES, **cases.SUPERSAMPLE_FRAME_CASES}.items():
        in_w, in_h, out_w, out_h = dims
        r = O.Ref(cases.make_ctx(ov))
        assert r.generateMapForPlane(*dims, 0)
        src = cases.case_input(name, in_w, in_h, pin)
        dst = np.full((out_h, out_w + pout), 0xA5, np.uint8)[:, :out_w]
        assert r.transformFramePlane(src, dst, 0)
        rec = {"out": hx(O.fnv1a64(np.ascontiguousarray(dst))), "in": hx(O.fnv1a64(np.ascontiguousarray(src)))}
        if cases.make_ctx(ov).enable_low_pass_filter:
            rec["blurred"] = hx(O.fnv1a64(r.filterPlane(src, 0)))
        frames[name] = rec
        r.close()

    here = os.path.dirname(os.path.abspath(__file__))
    for fname, obj in (("maps.json", maps), ("lowpass.json", lowpass), ("frames.json", frames)):
        with open(os.path.join(here, fname), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
            f.write("\n")
    print("wrote %d map, %d low-pass, %d frame vectors" % (len(maps), len(lowpass), len(frames)))


if __name__ == "__main__":
    main()
