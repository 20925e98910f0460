#!/usr/bin/env python3
"""Pin the oracle's restatement of OpenCV against REAL OpenCV.

The reference does its per-pixel arithmetic in three OpenCV calls -- cv::sepFilter2D
(/root/reference Transform360/Library/VideoFrameTransform.cpp:189-197), cv::remap (:748-754, :763-769) and
cv::resize(INTER_AREA) (:770-776).  OpenCV is an un-vendored, un-versioned dependency of the reference and is not
installed in the build image of this repository, so oracle/t360_oracle_cv.c restates its published algorithms and the
parity of that restatement is UNPINNED.  This script closes the gap on any machine that has `cv2` (pip install
opencv-python-headless) and `gcc` (for the oracle's own projection code, which IS pinned against the reference):

    python tests/golden/make_opencv_fixtures.py            # writes tests/golden/opencv_frames.npz (+ .json)
    python -m pytest tests/test_opencv_fixtures.py -q      # oracle vs OpenCV (CPU); with a GPU also the HIP path

For every case of tests/cases.py:OPENCV_CASES it takes
  * the warp map and the low-pass segments (rectangles + kernels) from the oracle's projection / filter-configuration
    code (bit-identical to the reference's own code: tests/test_oracle.py), and
  * the deterministic noise input of the case (tests/cases.py:case_input),
and restates VideoFrameTransform::transformPlane / filterPlane / filterSegment with cv2 calls only.  One thing differs
from the C++ text: a numpy slice handed to cv2 becomes a Mat that has forgotten its parent, so cv2.sepFilter2D would
replicate at the edge of the SEGMENT where the C++ ROI reads the real neighbours (replicating only at the edge of the
plane).  The script therefore filters the segment together with a margin of its real neighbours (of the plane padded
with BORDER_REPLICATE) and keeps the centre: the same pixels enter every output pixel as in the C++ call.

The committed file then lets tests/test_opencv_fixtures.py assert, without OpenCV: nearest = exact, everything else
within +-1 LSB (the bar BASELINE.json names), and it reports how many pixels differ at all.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def filter_plane_cv2(cv2, src, segs):
    """filterPlane (VideoFrameTransform.cpp:621-704, MONO) + filterSegment (:173-204): blurred = zeros, every segment
    sepFilter2D(in(rect), out(rect), -1, kX, kY, (-1,-1), 0, BORDER_REPLICATE) with the ROI's real neighbours."""
    h, w = src.shape
    blurred = np.zeros_like(src)
    for (left, top, sw, sh, kx, ky) in segs:
        if left < 0 or top < 0 or left + sw > w or top + sh > h or sw <= 0 or sh <= 0:
            continue  # cv::Mat::operator()(Rect) throws, filterSegment prints and returns
        rx, ry = len(kx) // 2, len(ky) // 2
        padded = cv2.copyMakeBorder(src, ry, ry, rx, rx, cv2.BORDER_REPLICATE)
        sub = np.ascontiguousarray(padded[top:top + sh + 2 * ry, left:left + sw + 2 * rx])
        out = cv2.sepFilter2D(sub, -1, np.asarray(kx, np.float32).reshape(1, -1), np.asarray(ky, np.float32).reshape(-1, 1),
                              anchor=(-1, -1), delta=0, borderType=cv2.BORDER_REPLICATE)
        blurred[top:top + sh, left:left + sw] = out[ry:ry + sh, rx:rx + sw]
    return blurred


def transform_plane_cv2(cv2, ctx, src, warp, segs, out_w, out_h, plane_index=0, prefill=0xA5):
    """transformPlane (VideoFrameTransform.cpp:707-794) for transformMatPlaneIndex 0."""
    from transform360_amd.abi import LAYOUT_BARREL, LAYOUT_BARREL_SPLIT
    barrel = ctx.output_layout in (LAYOUT_BARREL, LAYOUT_BARREL_SPLIT)
    border = cv2.BORDER_TRANSPARENT if barrel else cv2.BORDER_WRAP
    temp = filter_plane_cv2(cv2, src, segs) if ctx.enable_low_pass_filter else src
    mh, mw = warp.shape[:2]
    out = np.full((out_h, out_w), prefill, np.uint8)  # what the caller's buffer held (BORDER_TRANSPARENT keeps it)
    interp = int(ctx.interpolation_alg)  # the enum values ARE cv::INTER_* (Helper.h:49-54)
    if (out_h, out_w) == (mh, mw):
        if plane_index and barrel:
            out[:] = 128
        return cv2.remap(temp, warp, None, interp, dst=out, borderMode=border)  # dst is used in place (TRANSPARENT keeps it)
    scaled = np.full((mh, mw), 128 if plane_index else 0, np.uint8)
    scaled = cv2.remap(temp, warp, None, interp, dst=scaled, borderMode=border)
    return cv2.resize(scaled, (out_w, out_h), dst=out, fx=0, fy=0, interpolation=cv2.INTER_AREA)


def main():
    try:
        import cv2
    except ImportError:
        raise SystemExit("this script needs OpenCV's Python module (pip install opencv-python-headless); it is the one "
                         "thing the build image of this repository lacks")
    from oracle import t360_oracle as O
    from tests import cases
    O.build(ref=False)
    arrays, meta = {}, {"opencv_version": cv2.__version__, "cases": {}}
    for name, (ov, dims) in cases.OPENCV_CASES.items():
        in_w, in_h, out_w, out_h = dims
        ctx = cases.make_ctx(ov)
        o = O.Oracle(ctx)
        assert o.generateMapForPlane(in_w, in_h, out_w, out_h, 0), name
        warp = np.ascontiguousarray(o.map(0), dtype=np.float32)
        segs = o.segments(0) if ctx.enable_low_pass_filter else []
        src = np.ascontiguousarray(cases.case_input(name, in_w, in_h, 0))
        out = transform_plane_cv2(cv2, ctx, src, warp, segs, out_w, out_h)
        arrays[name] = out
        if ctx.enable_low_pass_filter:
            arrays[name + "__blurred"] = filter_plane_cv2(cv2, src, segs)
        # how the oracle compares, for the log
        want = np.full((out_h, out_w), 0xA5, np.uint8)
        assert o.transformFramePlane(src, want, 0)
        d = np.abs(out.astype(np.int16) - want.astype(np.int16))
        meta["cases"][name] = {"in": "%dx%d" % (in_w, in_h), "out": "%dx%d" % (out_w, out_h), "segments": len(segs),
                               "oracle_max_abs_diff": int(d.max()), "oracle_differing_pixels": int(np.count_nonzero(d))}
        print("%-32s OpenCV %s vs oracle: max |diff| %d, %d of %d pixels differ" % (name, cv2.__version__, d.max(),
                                                                                 np.count_nonzero(d), d.size))
        o.close()
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, "opencv_frames.npz"), **arrays)
    with open(os.path.join(here, "opencv_frames.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote tests/golden/opencv_frames.npz: %d arrays from OpenCV %s" % (len(arrays), cv2.__version__))


if __name__ == "__main__":
    main()
