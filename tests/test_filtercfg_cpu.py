"""CPU check of the library's HOST-side low-pass configuration (transform360_amd/csrc/t360_filtercfg.cpp is pure host
code; tests/plan_sim/filter_sim.cpp builds it with g++): segments, kernels and return values against the oracle for
configurations drawn from the whole space, and the shifted tap variants of the wide low-pass kernel against a direct
convolution."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.test_gpu_fuzz import draw
from transform360_amd.abi import FrameTransformContext, filter_defaults

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "plan_sim")


@pytest.fixture(scope="module")
def sim():
    so = os.path.join(SIM, "libfiltersim.so")
    srcs = [os.path.join(SIM, "filter_sim.cpp"), os.path.join(ROOT, "transform360_amd", "csrc", "t360_filtercfg.cpp")]
    deps = srcs + [os.path.join(ROOT, "transform360_amd", "csrc", n) for n in ("t360_filtercfg.h", "t360_internal.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "transform360_amd", "csrc")] + srcs + ["-o", so])
    L = C.CDLL(so)
    L.t360_host_filter_config.restype = C.c_int
    L.t360_host_filter_config.argtypes = [C.POINTER(FrameTransformContext)] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_int]
    L.t360_host_shifted_taps.restype = C.c_int
    L.t360_host_shifted_taps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return L


def _host_config(L, ctx, dims):
    cap, tap_cap = 4096, 1 << 20
    rects, lens = np.zeros(4 * cap, np.int32), np.zeros(2 * cap, np.int32)
    taps, q8 = np.zeros(tap_cap, np.float32), np.zeros(tap_cap, np.int32)
    # the library configures the filter for the SCALED output size (reference :560-565)
    sw = int(ctx.width_scale_factor * dims[2] + 0.5)
    sh = int(ctx.height_scale_factor * dims[3] + 0.5)
    n = L.t360_host_filter_config(C.byref(ctx), dims[0], dims[1], sw, sh, cap, rects.ctypes.data, lens.ctypes.data,
                                  taps.ctypes.data, q8.ctypes.data, tap_cap)
    return n, rects, lens, taps


@pytest.mark.parametrize("seed", range(200))
def test_host_filter_config_equals_oracle(seed, sim, oracle_mod):
    O = oracle_mod
    ov, dims, _, _ = draw(20000 + seed)
    ov["enable_low_pass_filter"] = 1
    ov.setdefault("num_vertical_segments", 1 + seed % 7)
    ov.setdefault("num_horizontal_segments", 1 + seed % 4)
    ctx = filter_defaults(**ov)
    o = O.Oracle(ctx)
    ok_o = bool(o.generateMapForPlane(*dims, 0))
    n, rects, lens, taps = _host_config(sim, ctx, dims)
    assert n != -2
    assert ok_o == (n >= 0), "oracle %s, library's host configuration %d for %r %r" % (ok_o, n, ov, dims)
    if not ok_o:
        return
    segs = o.segments(0)
    assert len(segs) == n
    at = 0
    for i, s in enumerate(segs):   # (left, top, width, height, kx, ky)
        assert tuple(s[:4]) == tuple(int(v) for v in rects[4 * i:4 * i + 4])
        assert (len(s[4]), len(s[5])) == (int(lens[2 * i]), int(lens[2 * i + 1]))
        for k in (s[4], s[5]):
            got = taps[at:at + len(k)]
            assert np.array_equal(np.asarray(k, np.float32).view(np.uint32), got.view(np.uint32))
            at += len(k)


@pytest.mark.parametrize("ntaps", [1, 3, 5, 7, 9, 11, 13, 15, 21, 35, 37, 38, 39, 41])
def test_shifted_tap_variants_are_the_convolution(ntaps, sim):
    rng = np.random.default_rng(ntaps)
    kx = rng.integers(0, 256, ntaps).astype(np.int32)
    out = np.zeros(4 * 12, np.uint32)
    nd = sim.t360_host_shifted_taps(kx.ctypes.data, ntaps, out.ctypes.data)
    rx, m = ntaps // 2, (4 - (ntaps // 2) % 4) % 4
    if (ntaps + m + 6) // 4 > 11:
        assert nd == 0
        return
    assert nd == (ntaps + m + 6) // 4
    row = rng.integers(0, 256, 256).astype(np.int64)
    var = out.view(np.uint8).reshape(4, 48).astype(np.int64)
    assert not var[:, 4 * nd:].any()   # nothing beyond the nd dwords the kernel reads
    for px0 in (64, 68, 100):          # output pixels px0 + j, px0 % 4 == 0
        base = px0 - rx - m            # the aligned byte the lane's window starts at
        assert base % 4 == 0
        for j in range(4):
            dot = int((row[base:base + 48] * var[j]).sum())
            want = int((row[px0 + j - rx:px0 + j - rx + ntaps] * kx).sum())
            assert dot == want
