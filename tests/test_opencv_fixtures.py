"""Oracle (and, on a GPU, the HIP path) against outputs of REAL OpenCV -- when somebody has produced them.

tests/golden/opencv_frames.npz is written by tests/golden/make_opencv_fixtures.py on a machine that has cv2 (the
build image of this repository does not: OpenCV is an un-vendored dependency of the reference).  Until the file is
committed these tests SKIP and the parity of oracle/t360_oracle_cv.c stays "unpinned"; once it is there they assert
the bar of BASELINE.json: nearest bit-exact, every other interpolation within +-1 LSB per 8-bit sample.
"""
import os

import numpy as np
import pytest

from tests import cases

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = os.path.join(HERE, "golden", "opencv_frames.npz")

needs_fixture = pytest.mark.skipif(
    not os.path.exists(NPZ),
    reason="tests/golden/opencv_frames.npz not generated yet: run tests/golden/make_opencv_fixtures.py where cv2 is installed")


def _tolerance(ov):
    from transform360_amd.abi import NEAREST
    ctx = cases.make_ctx(ov)
    exact = int(ctx.interpolation_alg) == NEAREST and not ctx.enable_low_pass_filter and \
        ctx.width_scale_factor == 1.0 and ctx.height_scale_factor == 1.0
    return 0 if exact else 1


def test_recipe_cases_are_runnable_without_opencv():
    """The recipe's inputs (maps, segments, noise) come from code that runs here: every case initialises, and the cv2-free
    half of the script -- the oracle's own output for the case -- is produced (keeps the recipe from rotting)."""
    from oracle import t360_oracle as O
    for name, (ov, dims) in cases.OPENCV_CASES.items():
        in_w, in_h, out_w, out_h = dims
        o = O.Oracle(cases.make_ctx(ov))
        assert o.generateMapForPlane(in_w, in_h, out_w, out_h, 0), name
        src = cases.case_input(name, in_w, in_h, 0)
        want = np.full((out_h, out_w), 0xA5, np.uint8)
        assert o.transformFramePlane(src, want, 0), name
        o.close()


@needs_fixture
@pytest.mark.parametrize("name", sorted(cases.OPENCV_CASES))
def test_oracle_matches_opencv(name):
    from oracle import t360_oracle as O
    ov, (in_w, in_h, out_w, out_h) = cases.OPENCV_CASES[name]
    got_cv = np.load(NPZ)[name]
    o = O.Oracle(cases.make_ctx(ov))
    assert o.generateMapForPlane(in_w, in_h, out_w, out_h, 0)
    src = cases.case_input(name, in_w, in_h, 0)
    want = np.full((out_h, out_w), 0xA5, np.uint8)
    assert o.transformFramePlane(src, want, 0)
    d = np.abs(got_cv.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= _tolerance(ov), "%s: oracle differs from OpenCV by %d (%d pixels differ)" % (name, d.max(), np.count_nonzero(d))
    if cases.make_ctx(ov).enable_low_pass_filter:
        b = np.abs(np.load(NPZ)[name + "__blurred"].astype(np.int16) - o.filterPlane(src, 0).astype(np.int16))
        assert b.max() <= 1, "%s: low-pass differs from OpenCV by %d" % (name, b.max())
    o.close()


@needs_fixture
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.OPENCV_CASES))
def test_hip_path_matches_opencv(name):
    import torch

    from transform360_amd import handler as T
    ov, (in_w, in_h, out_w, out_h) = cases.OPENCV_CASES[name]
    got_cv = np.load(NPZ)[name]
    with T.VideoFrameTransform(cases.make_ctx(ov)) as t:
        assert t.generateMapForPlane(in_w, in_h, out_w, out_h, 0)
        src = torch.from_numpy(np.ascontiguousarray(cases.case_input(name, in_w, in_h, 0))).cuda()
        dst = torch.full((out_h, out_w), 0xA5, dtype=torch.uint8, device="cuda")
        assert t.transformFramePlane(src, dst, 0)
        torch.cuda.synchronize()
        d = np.abs(got_cv.astype(np.int16) - dst.cpu().numpy().astype(np.int16))
    assert d.max() <= _tolerance(ov), "%s: HIP path differs from OpenCV by %d (%d pixels differ)" % (name, d.max(), np.count_nonzero(d))
