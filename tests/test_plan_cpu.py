"""CPU check of the gather planner (transform360_amd/csrc/t360_plan.cpp is pure host code): tests/plan_sim builds it
with g++ and EMULATES the gather through the plan -- every tile's chunk table is staged into a fake LDS, every pixel's
stencil rows are looked up the way the kernel does (pixel word -> row table -> LDS address) and the bytes found there are
compared with the source sampled directly from the LUT; every output pixel must be covered exactly once.  The LUT is the
oracle's map quantised the way cv::remap does."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "plan_sim"))

from transform360_amd.abi import (CUBIC, LANCZOS4, LAYOUT_CUBEMAP_23_OFFCENTER, LAYOUT_EAC_32, LINEAR, NEAREST,  # noqa: E402
                                  STEREO_FORMAT_TB, filter_defaults)

CASES = {
    "cube_bicubic": (dict(interpolation_alg=CUBIC), (640, 320, 384, 256)),
    "cube_nearest": (dict(interpolation_alg=NEAREST), (640, 320, 384, 256)),
    "cube_bilinear_rotated": (dict(interpolation_alg=LINEAR, fixed_yaw=33.0, fixed_pitch=-21.0, fixed_roll=9.0), (512, 256, 288, 192)),
    "cube_lanczos": (dict(interpolation_alg=LANCZOS4), (512, 256, 192, 128)),
    "cube23_offcentre": (dict(interpolation_alg=CUBIC, output_layout=LAYOUT_CUBEMAP_23_OFFCENTER, fixed_cube_offcenter_z=0.4),
                         (640, 320, 256, 384)),
    "eac_tb": (dict(interpolation_alg=CUBIC, output_layout=LAYOUT_EAC_32, input_stereo_format=STEREO_FORMAT_TB,
                    output_stereo_format=STEREO_FORMAT_TB), (512, 512, 288, 384)),
    "ragged_output": (dict(interpolation_alg=CUBIC), (640, 320, 300, 200)),
    "config1_luma": (dict(interpolation_alg=NEAREST), (1920, 960, 768, 512)),      # BASELINE config 1
    "config2_luma": (dict(interpolation_alg=CUBIC), (3840, 1920, 1536, 1024)),     # BASELINE config 2, both plane shapes
    "config2_chroma": (dict(interpolation_alg=CUBIC), (1920, 960, 768, 512)),
}


@pytest.fixture(scope="module")
def sim():
    import plan_sim
    L = plan_sim.build()
    L.t360_plan_verify.restype = C.c_longlong
    L.t360_plan_verify.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
    return L


@pytest.mark.parametrize("waves,pieces", [(8, 24), (4, 16), (8 | (1000 << 8) | (1 << 20), 24),   # the third: 256x8 tiles wherever they fit
                                          (8 | (4 << 24), 24), (8 | (2 << 24), 24),                 # scatter tiles, strips of 4 / 2 lines
                                          (4 | (1 << 20), 12),    # shapes compared by lines on 4 waves: what nearest maps run with
                                          (8 | (1 << 20), 24),    # ... and on 8 waves: the fallback of a nearest map the 4-wave planner refuses
                                          (8 | (4 << 24), 1)])    # scatter groups that cannot be staged go back to rectangles
@pytest.mark.parametrize("name", sorted(CASES))
def test_gather_through_the_plan_reads_the_right_bytes(name, waves, pieces, sim, oracle_mod):
    O = oracle_mod
    ov, (in_w, in_h, out_w, out_h) = CASES[name]
    ctx = filter_defaults(enable_low_pass_filter=0, **ov)
    o = O.Oracle(ctx, threads=4)
    assert o.generateMapForPlane(in_w, in_h, out_w, out_h, 0)
    q, nn = O.quantize_map(o.map(0))
    ks = {NEAREST: 1, LINEAR: 2, CUBIC: 4, LANCZOS4: 8}[int(ctx.interpolation_alg)]
    lut = np.zeros(q.shape[:2] + (4,), np.int16)
    if ks == 1:
        lut[..., 0], lut[..., 1] = np.clip(nn[..., 0], -32768, 32767), np.clip(nn[..., 1], -32768, 32767)
    else:
        lut[..., 0], lut[..., 1] = np.clip(q[..., 0], -32768, 32767), np.clip(q[..., 1], -32768, 32767)
        lut[..., 2] = q[..., 2].astype(np.int16)
    lut = np.ascontiguousarray(lut)
    src = np.random.default_rng(3).integers(0, 256, (in_h, in_w), dtype=np.uint8)
    if ks == 8:
        waves, pieces = 4, min(pieces, 16)   # Lanczos4 plans: workgroups of 4 waves, 16x16 tiles
    bad = sim.t360_plan_verify(lut.ctypes.data, out_w, out_h, in_w, in_h, ks, pieces, waves, src.ctypes.data)
    assert bad == 0


@pytest.mark.parametrize("interp,ks", [(LINEAR, 2), (CUBIC, 4), (LANCZOS4, 8)])
def test_packed_weights_reproduce_the_q15_table(interp, ks, sim, oracle_mod):
    """pack_weights(): w = 256 * (signed high byte) + (unsigned low byte), window (r, q) = taps 4q..4q+3 of stencil row r,
    for every phase of OpenCV's table."""
    tab = oracle_mod.inter_tab(interp).astype(np.int16)           # [1024, ks * ks]
    stride = {2: 4, 4: 8, 8: 32}[ks]
    out = np.zeros(1024 * stride, np.uint32)
    sim.t360_pack_weights.restype = C.c_int
    sim.t360_pack_weights.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    assert sim.t360_pack_weights(np.ascontiguousarray(tab).ctypes.data, ks, out.ctypes.data) == out.size
    win = 2 if ks == 8 else 1
    nw = ks * win
    o = out.reshape(1024, stride)
    hi = o[:, :nw].copy().view(np.int8).reshape(1024, nw, 4).astype(np.int64)
    lo = o[:, nw:2 * nw].copy().view(np.uint8).reshape(1024, nw, 4).astype(np.int64)
    w = (256 * hi + lo).reshape(1024, ks, win * 4)[:, :, :ks].reshape(1024, ks * ks)
    assert np.array_equal(w, tab.astype(np.int64))
    unused = (256 * hi + lo).reshape(1024, ks, win * 4)[:, :, ks:]
    assert not unused.any()                                        # bilinear: bytes 2-3 of its window
    assert stride == 2 * nw                                        # nothing but the two halves (the kernel derives the bias)


FUSED_CASES = {
    "config3_luma": (dict(num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1), CUBIC, (3840, 1920, 1536, 1024)),
    "config3_chroma": (dict(num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1), CUBIC, (1920, 960, 768, 512)),
    "filter_defaults_5x1": (dict(), CUBIC, (1280, 640, 768, 512)),
    "three_bands_per_tile": (dict(num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1), CUBIC, (480, 240, 192, 128)),
    "bilinear_9x4": (dict(num_vertical_segments=9, num_horizontal_segments=4), LINEAR, (960, 480, 384, 256)),
    "rotated_ragged": (dict(num_vertical_segments=15, num_horizontal_segments=8, fixed_yaw=33.0, fixed_pitch=-21.0), CUBIC, (640, 320, 300, 200)),
}


@pytest.mark.parametrize("name", sorted(FUSED_CASES))
def test_fused_lowpass_tiles_filter_and_gather_the_right_bytes(name, sim, oracle_mod):
    """The fused low-pass work list of a plan (t360_internal.h "fused low-pass tiles"), emulated on the CPU: every fused tile's
    RAW footprint is staged through its R chunk table, every lane's run is filtered with the kernel's integer arithmetic and
    written in place, and the bytes under every pixel's stencil are compared with the oracle's filtered plane; the plan's
    UNFUSED tiles are staged from a filtered plane in which every segment the plan does not list as needed is inverted; every
    output pixel is covered exactly once."""
    O = oracle_mod
    ov, interp, (in_w, in_h, out_w, out_h) = FUSED_CASES[name]
    ctx = filter_defaults(interpolation_alg=interp, enable_low_pass_filter=1, **ov)
    o = O.Oracle(ctx, threads=4)
    assert o.generateMapForPlane(in_w, in_h, out_w, out_h, 0)
    q, _ = O.quantize_map(o.map(0))
    lut = np.zeros(q.shape[:2] + (4,), np.int16)
    lut[..., 0], lut[..., 1] = np.clip(q[..., 0], -32768, 32767), np.clip(q[..., 1], -32768, 32767)
    lut[..., 2] = q[..., 2].astype(np.int16)
    lut = np.ascontiguousarray(lut)
    src = np.random.default_rng(5).integers(0, 256, (in_h, in_w), dtype=np.uint8)
    blurred = o.filterPlane(src, 0)
    row_kid, rects, taps, nsegs = np.zeros(in_h, np.int16), np.zeros(4 * 4096, np.int32), np.zeros(16 * 512, np.uint32), C.c_int()
    sim.t360_host_fuse_info.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    nk = sim.t360_host_fuse_info(C.byref(ctx), in_w, in_h, out_w, out_h, row_kid.ctypes.data, rects.ctypes.data, 4096,
                                 taps.ctypes.data, 512, C.byref(nsegs))
    assert nk > 0 and (row_kid >= 0).any()
    sim.t360_plan_verify_fused.restype = C.c_longlong
    sim.t360_plan_verify_fused.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                                         C.c_void_p, C.c_void_p]
    st = (C.c_longlong * 8)()
    ks = {LINEAR: 2, CUBIC: 4}[interp]
    bad = sim.t360_plan_verify_fused(lut.ctypes.data, out_w, out_h, in_w, in_h, ks, 24, src.ctypes.data, blurred.ctypes.data,
                                     rects.ctypes.data, nsegs.value, row_kid.ctypes.data, taps.ctypes.data, st)
    assert bad == 0
    assert st[0] > 0                       # something was fused ...
    assert st[3] < nsegs.value or st[1] + st[2] == 0 or name == "three_bands_per_tile"   # ... and fewer segments are needed


def test_request_model_matches_the_measured_l1_requests(oracle_mod):
    """tests/plan_sim/request_model.py: the distinct 128-byte lines under every 1 KiB DMA instruction
    of the shipped plan, summed over the planes of BASELINE config 2 and 64 frames, against TCP_TCC_READ_REQ of the profiled
    launch (profiles/r05_pmc_summary.txt: 14.18 M; the remainder is the pole tiles' direct loads and the per-tile tables)."""
    import plan_sim
    import request_model
    L = plan_sim.build()
    req = pieces = 0
    for plane, copies in ((0, 1), (1, 2)):
        rows, _ = request_model.model(L, 2, plane, 24, 8)
        req += copies * rows[8][2]
        pieces += copies * rows[8][1]
    assert 0.88 <= req * 64 / 14176454.1 <= 1.0
    assert 13.0 <= req / pieces <= 14.5          # requests per staged KiB (whole-line staging would need 8)
