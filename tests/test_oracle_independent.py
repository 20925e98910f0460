"""Independent cross-checks of oracle/t360_oracle_cv.c (CPU only, no OpenCV in this image).

The oracle's restatement of cv::remap / cv::sepFilter2D / cv::resize(INTER_AREA) is "parity unpinned": no OpenCV-made
vector exists here (tests/golden/make_opencv_fixtures.py is the recipe for a machine that has cv2).  What CAN be done
without OpenCV is to check the restatement against the MATHEMATICAL DEFINITION of each operation, written a second time
in numpy / scipy / PIL with none of the oracle's tables or code:

  * remap: float64 bilinear, bicubic (A = -0.75) and direct-definition Lanczos4 (sinc(t) sinc(t/4), normalised) on the
    1/32-pixel-quantised coordinates OpenCV uses, BORDER_WRAP in both axes -> within +-1 LSB; nearest -> exact;
  * sepFilter2D, fixed-point path: scipy.ndimage.correlate1d with the Q8-rounded integer kernels, replicate border at
    the parent's edges and REAL neighbours at a ROI's edges, (sum + 2^15) >> 16 -> exact; float path -> within +-1;
  * INTER_AREA: block mean for integer factors (2x2 exact, others +-1), an exact area-weighted average in float64 for
    fractional factors (+-0.5; PIL's BOX filter was tried as a second opinion and is NOT an area average for fractional
    factors -- it counts whole source pixels by their centres), OpenCV's area-mode linear interpolation for enlarging factors (+-1).

These catch a misremembered constant, tap order, rounding rule or border rule -- which oracle-vs-itself cannot.  They do
NOT pin OpenCV's exact rounding on ties; SURVEY 8c stays "partial" until an OpenCV-made fixture is committed.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import t360_oracle as O

NEAREST, LINEAR, CUBIC, LANCZOS4 = 0, 1, 2, 4   # cv::InterpolationFlags, the values the reference passes through
BORDER_WRAP = 3


# ---------------------------------------------------------------- remap ------------------------------------------------

def w_linear(f):
    return np.stack([1.0 - f, f], axis=-1)


def w_cubic(f, A=-0.75):
    def near(t):   # |t| <= 1
        return ((A + 2.0) * t - (A + 3.0)) * t * t + 1.0

    def far(t):    # 1 < |t| < 2
        return ((A * t - 5.0 * A) * t + 8.0 * A) * t - 4.0 * A
    return np.stack([far(1.0 + f), near(f), near(1.0 - f), far(2.0 - f)], axis=-1)


def w_lanczos4(f):
    # taps at integer offsets -3..4 from floor(x); t = offset - f; w = sinc(t) sinc(t/4), then normalised to sum 1
    offs = np.arange(-3, 5, dtype=np.float64)
    t = offs[None, :] - f[:, None]
    w = np.sinc(t) * np.sinc(t / 4.0)
    return w / w.sum(axis=-1, keepdims=True)


def remap_definition(src, mapxy, interp):
    """cv::remap by its definition: coordinates rounded to 1/32 px (imgwarp.cpp: cvRound(x * INTER_TAB_SIZE)), separable
    weights of the fractional part, BORDER_WRAP; float64 accumulation, round half to even at the end."""
    sh, sw = src.shape
    x = mapxy[..., 0].astype(np.float32).ravel()
    y = mapxy[..., 1].astype(np.float32).ravel()
    if interp == NEAREST:
        ix = np.rint(x).astype(np.int64) % sw
        iy = np.rint(y).astype(np.int64) % sh
        return src[iy, ix].reshape(mapxy.shape[:2])
    sx = np.rint(x * np.float32(32)).astype(np.int64)
    sy = np.rint(y * np.float32(32)).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    fx, fy = (sx & 31) / 32.0, (sy & 31) / 32.0
    wf, lo = {LINEAR: (w_linear, 0), CUBIC: (w_cubic, 1), LANCZOS4: (w_lanczos4, 3)}[interp]
    wx, wy = wf(fx), wf(fy)
    k = wx.shape[1]
    acc = np.zeros(x.shape, np.float64)
    s = src.astype(np.float64)
    for r in range(k):
        yy = (iy - lo + r) % sh
        for c in range(k):
            xx = (ix - lo + c) % sw
            acc += wy[:, r] * wx[:, c] * s[yy, xx]
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8).reshape(mapxy.shape[:2])


def oracle_remap(src, mapxy, interp):
    L = O.lib()
    dh, dw = mapxy.shape[:2]
    dst = np.zeros((dh, dw), np.uint8)
    m = np.ascontiguousarray(mapxy, np.float32)
    L.t360o_remap_rows(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0], dst.ctypes.data, dw, dh, dst.strides[0],
                       m.ctypes.data, interp, BORDER_WRAP, 0, dh)
    return dst


def random_map(rng, sw, sh, dw, dh, smooth):
    if smooth:
        # an equirect-like warp: smooth, magnifying and minifying regions, crosses the wrap seam and both poles
        u, v = np.meshgrid(np.linspace(0, 1, dw), np.linspace(0, 1, dh))
        x = (u * 1.3 - 0.15) * sw + 40 * np.sin(v * 7.0)
        y = (v * 1.2 - 0.1) * sh + 25 * np.cos(u * 5.0)
    else:
        x = rng.uniform(-3.0, sw + 3.0, (dh, dw))
        y = rng.uniform(-3.0, sh + 3.0, (dh, dw))
    return np.stack([x, y], axis=-1).astype(np.float32)


@pytest.mark.parametrize("interp", [NEAREST, LINEAR, CUBIC, LANCZOS4])
@pytest.mark.parametrize("smooth", [True, False])
def test_remap_matches_its_float_definition(interp, smooth):
    rng = np.random.default_rng(1000 * interp + smooth)
    sw, sh, dw, dh = 1024, 512, 384, 256
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    m = random_map(rng, sw, sh, dw, dh, smooth)
    got = oracle_remap(src, m, interp)
    want = remap_definition(src, m, interp)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    if interp == NEAREST:
        assert d.max() == 0
    else:
        assert d.max() <= 1, "max |diff| %d" % d.max()
        # the integer formulation differs from exact arithmetic only on near-ties: a few percent at most
        assert np.count_nonzero(d) < 0.03 * d.size, "%.2f %% of the pixels differ" % (100.0 * np.count_nonzero(d) / d.size)


def test_remap_on_a_gradient_is_unbiased():
    """a linear ramp is reproduced exactly by every interpolating kernel whose weights sum to 1 with first moment f
    (catches a shifted tap origin: off by one tap = off by the ramp's slope)"""
    sw, sh = 512, 256
    src = np.tile((np.arange(sw) // 2).astype(np.uint8), (sh, 1))   # slope 0.5 per px, no wrap inside the sampled range
    u, v = np.meshgrid(np.linspace(20.25, sw - 20.75, 300), np.linspace(10.5, sh - 10.5, 40))
    m = np.stack([u, v], axis=-1).astype(np.float32)
    for interp in (LINEAR, CUBIC, LANCZOS4):
        got = oracle_remap(src, m, interp).astype(np.float64)
        ideal = np.rint(m[..., 0].astype(np.float64) * 32) / 32 / 2.0   # the ramp at the quantised coordinate
        # the source is the FLOOR of the ramp (steps of 1 every 2 px): allow for that staircase, not for a tap shift
        assert np.abs(got - ideal).max() <= 1.0, interp


# ---------------------------------------------------------------- sepFilter2D ------------------------------------------

def gaussian(sigma, n):
    xs = np.arange(n) - (n - 1) / 2.0
    k = np.exp(-xs * xs / (2 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def oracle_sepfilter(parent, roi, kx, ky):
    L = O.lib()
    out = parent.copy()
    left, top, w, h = roi
    path = L.t360o_sepfilter_roi(parent.ctypes.data, parent.shape[1], parent.shape[0], parent.strides[0], out.ctypes.data,
                                 out.strides[0], left, top, w, h, kx.ctypes.data, len(kx), ky.ctypes.data, len(ky))
    return path, out


@pytest.mark.parametrize("kxn,kyn,sx,sy", [(3, 3, 0.8, 0.6), (5, 3, 1.1, 0.7), (13, 5, 2.6, 1.0), (31, 15, 6.0, 3.0), (1, 3, 1.0, 0.9)])
def test_sepfilter_fixed_point_path_is_an_integer_correlation(kxn, kyn, sx, sy):
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(kxn * 100 + kyn)
    parent = rng.integers(0, 256, (256, 512), dtype=np.uint8)
    kx = gaussian(sx, kxn) if kxn > 1 else np.ones(1, np.float32)
    ky = gaussian(sy, kyn)
    # Q8 kernels as OpenCV's fixed-point separable filter makes them (kernel * 256, rounded to nearest)
    kx8 = np.rint(kx.astype(np.float64) * 256).astype(np.int64)
    ky8 = np.rint(ky.astype(np.float64) * 256).astype(np.int64)
    rows = correlate1d(parent.astype(np.int64), kx8, axis=1, mode="nearest")
    full = (correlate1d(rows, ky8, axis=0, mode="nearest") + (1 << 15)) >> 16
    full = np.clip(full, 0, 255).astype(np.uint8)
    for roi in ((0, 0, 512, 256), (64, 32, 128, 64), (0, 100, 40, 156), (500, 0, 12, 20)):
        path, got = oracle_sepfilter(parent, roi, kx, ky)
        l, t, w, h = roi
        want = parent.copy()
        want[t:t + h, l:l + w] = full[t:t + h, l:l + w]   # a ROI reads its real neighbours, the parent's edge replicates
        if kxn == 1:
            # a one-tap kernel [1.0] is KERNEL_INTEGER as well as smooth + symmetric, and cv::getKernelType's result
            # must EQUAL smooth + symmetric for the 8-bit fixed-point filter: OpenCV (and the oracle) go through float
            assert path == 0
            assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 1
            continue
        assert path == 1, "expected the fixed-point path for a normalised symmetric kernel"
        assert np.array_equal(got, want), "roi %s: %d pixels differ" % (roi, np.count_nonzero(got != want))


def test_sepfilter_float_path_within_one_lsb():
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(7)
    parent = rng.integers(0, 256, (128, 256), dtype=np.uint8)
    kx = np.array([0.1, 0.2, 0.5, 0.15, 0.05], np.float32)   # asymmetric: OpenCV's fixed-point path needs symmetry
    ky = np.array([0.25, 0.45, 0.3], np.float32)
    path, got = oracle_sepfilter(parent, (0, 0, 256, 128), kx, ky)
    assert path == 0, "an asymmetric kernel must take the float path"
    rows = correlate1d(parent.astype(np.float64), kx.astype(np.float64), axis=1, mode="nearest")
    want = np.clip(np.rint(correlate1d(rows, ky.astype(np.float64), axis=0, mode="nearest")), 0, 255)
    assert np.abs(got.astype(np.float64) - want).max() <= 1


# ---------------------------------------------------------------- INTER_AREA -------------------------------------------

def oracle_resize(src, dw, dh):
    L = O.lib()
    L.t360o_resize_area.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
    dst = np.zeros((dh, dw), np.uint8)
    assert L.t360o_resize_area(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0], dst.ctypes.data, dw, dh, dst.strides[0])
    return dst


def area_weights(n_src, n_dst):
    """row-stochastic matrix of the exact area overlap: dst cell d covers [d*s, (d+1)*s) of the source axis"""
    s = n_src / n_dst
    W = np.zeros((n_dst, n_src))
    for d in range(n_dst):
        a, b = d * s, (d + 1) * s
        for i in range(int(np.floor(a)), min(n_src, int(np.ceil(b)))):
            W[d, i] = max(0.0, min(b, i + 1) - max(a, i))
    return W / W.sum(axis=1, keepdims=True)


@pytest.mark.parametrize("fx,fy", [(2, 2), (3, 2), (3, 3), (4, 1), (1, 2)])
def test_inter_area_integer_factors_are_block_means(fx, fy):
    rng = np.random.default_rng(fx * 10 + fy)
    dw, dh = 96, 64
    src = rng.integers(0, 256, (dh * fy, dw * fx), dtype=np.uint8)
    got = oracle_resize(src, dw, dh)
    blocks = src.reshape(dh, fy, dw, fx).astype(np.int64).sum(axis=(1, 3))
    if (fx, fy) == (2, 2):
        assert np.array_equal(got, ((blocks + 2) >> 2).astype(np.uint8))   # OpenCV's 2x2 fast path: integer, half up
    else:
        mean = blocks / float(fx * fy)
        assert np.abs(got.astype(np.float64) - mean).max() <= 0.5 + 1e-3   # a correctly rounded mean


@pytest.mark.parametrize("sw,sh,dw,dh", [(300, 200, 200, 160), (257, 129, 100, 50), (512, 96, 360, 40), (120, 90, 84, 36)])
def test_inter_area_fractional_factors_are_area_averages(sw, sh, dw, dh):
    rng = np.random.default_rng(sw + dh)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    got = oracle_resize(src, dw, dh).astype(np.float64)
    exact = area_weights(sh, dh) @ src.astype(np.float64) @ area_weights(sw, dw).T
    assert np.abs(got - exact).max() <= 0.5 + 1e-3, "not the area-weighted average"


@pytest.mark.parametrize("sw,sh,dw,dh", [(100, 80, 150, 120), (64, 64, 128, 128), (90, 50, 200, 75)])
def test_inter_area_enlarging_is_area_mode_linear_interpolation(sw, sh, dw, dh):
    """cv::resize with INTER_AREA and a factor that enlarges falls back to linear interpolation with the AREA-mode
    coefficients (resize.cpp): sx = floor(dx * scale), fx = (dx + 1) - (sx + 1) * inv_scale, clamped to [0, 1)."""
    rng = np.random.default_rng(sw * dw)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8).astype(np.float64)

    def coeffs(n_src, n_dst):
        inv = n_dst / n_src
        scale = 1.0 / inv          # resize.cpp computes the scale as the reciprocal of the double ratio, not as n_src / n_dst:
        d = np.arange(n_dst)      # floor(dx * scale) differs between the two at exact multiples (90 -> 200: dx = 20, 40, ...)
        s0 = np.floor(d * scale).astype(np.int64)
        f = (d + 1) - (s0 + 1) * inv
        f = np.where(f <= 0, 0.0, f - np.floor(f))
        s1 = np.minimum(s0 + 1, n_src - 1)
        f = np.where(s0 >= n_src - 1, 0.0, f)
        s0 = np.minimum(s0, n_src - 1)
        return s0, s1, f

    if dw < sw or dh < sh:
        pytest.skip("mixed shrink / enlarge: covered by the GPU fuzz against the oracle only")
    x0, x1, fx = coeffs(sw, dw)
    y0, y1, fy = coeffs(sh, dh)
    rows = src[:, x0] * (1 - fx) + src[:, x1] * fx
    want = rows[y0, :] * (1 - fy)[:, None] + rows[y1, :] * fy[:, None]
    got = oracle_resize(src.astype(np.uint8), dw, dh).astype(np.float64)
    assert np.abs(got - want).max() <= 1.0, "max diff %.2f" % np.abs(got - want).max()


# ---------------------------------------------------------------- the Q15 coefficient tables ---------------------------

@pytest.mark.parametrize("interp,wf", [(LINEAR, w_linear), (CUBIC, w_cubic), (LANCZOS4, w_lanczos4)])
def test_q15_tables_are_the_rounded_separable_weights(interp, wf):
    """initInterTab2D: entry (fy, fx) holds round(wy[r] * wx[c] * 32768) with the rounding error of the 16..64 taps pushed
    into the central taps so that every entry sums to exactly 32768."""
    tab = O.inter_tab(interp).astype(np.int64)          # [1024, k*k]
    k = int(round(tab.shape[1] ** 0.5))
    assert np.all(tab.sum(axis=1) == 32768)
    f = np.arange(32) / 32.0
    w1 = wf(f)                                          # [32, k]
    ideal = (w1[:, None, :, None] * w1[None, :, None, :]).reshape(1024, k * k) * 32768.0   # index = fy * 32 + fx
    d = np.abs(tab - ideal)
    # The sum fix-up moves the accumulated rounding error (a few units) into ONE entry of rows / columns k/2 .. k/2+1 of
    # the stencil -- OpenCV's scan starts at ksize/2, so for bicubic that is the lower-right 2x2 block (rows 2..3), not the
    # block of the four largest weights; bilinear products are exact and never need it.
    fix = np.zeros((k, k), bool)
    fix[k // 2:k // 2 + 2, k // 2:k // 2 + 2] = True
    fix = fix.ravel()
    # a weight of exactly 1.0 (phase 0, and the pure-x / pure-y phases' products with it) is 32768 = one more than a
    # short holds: saturate_cast<short> stores 32767 and the fix-up puts the missing 1 into its own 2x2 block -- for
    # bilinear that block lies outside the 2x2 stencil's first entry too (entry [1][1] of phase 0 becomes 1)
    sat = ideal > 32767.5
    assert np.all(tab[sat] == 32767)
    d = np.where(sat, 0.0, d)
    assert d[:, ~fix].max() <= 0.51           # float32 products: 0.5 + 32768 * 2^-24 * a few
    assert d[:, fix].max() <= 0.51 + k * k / 2.0
    assert (d[:, fix] > 0.51).sum(axis=1).max() <= 1   # one entry per phase absorbs the difference
