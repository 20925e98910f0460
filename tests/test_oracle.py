"""CPU suite: the oracle against the committed golden vectors (tests/golden/make_golden.py) and -- where
/root/reference is mounted -- against the reference build itself (oracle/_ref), entry by entry.

Provenance of the vectors: maps.json and lowpass.json come from the reference's own projection and filter-config
code; frames.json is SHIM-GENERATED -- the reference's frame path ran, but its cv::remap / cv::sepFilter2D /
cv::resize calls were bound to the oracle's restatement of OpenCV, so those vectors pin the orchestration (segments,
eyes, border modes, resize branch) and NOT OpenCV's arithmetic (parity unpinned at that boundary)."""
import numpy as np
import pytest

from tests import cases
from transform360_amd.abi import CUBIC, LANCZOS4, LINEAR, NEAREST, filter_defaults

ALL_MAPS = {**cases.MAP_CASES, **cases.LAYOUT_MAP_CASES}
ALL_FRAMES = {**cases.FRAME_CASES, **cases.LAYOUT_FRAME_CASES, **cases.SUPERSAMPLE_FRAME_CASES}
BIG = ("cfg4_luma", "cfg4_chroma")


def hx(v):
    return "%016x" % v


@pytest.mark.parametrize("name", sorted(ALL_MAPS))
def test_map_matches_golden(name, oracle_mod, golden):
    O = oracle_mod
    ov, dims = ALL_MAPS[name]
    o = O.Oracle(cases.make_ctx(ov))
    assert o.generateMapForPlane(*dims, 0)
    m = o.map(0)
    g = golden["maps"][name]
    assert m.shape == (g["h"], g["w"], 2)
    q, nn = O.quantize_map(m)
    assert hx(O.fnv1a64(m)) == g["f32"]          # float bits of every coordinate
    assert hx(O.fnv1a64(q)) == g["q"]            # 1/32-px quanta (bilinear/bicubic/lanczos)
    assert hx(O.fnv1a64(nn)) == g["nn"]          # nearest picks
    for r, c, xh, yh in g["samples"]:
        assert float(m[r, c, 0]).hex() == xh and float(m[r, c, 1]).hex() == yh


@pytest.mark.parametrize("name", sorted(n for n in ALL_MAPS if n not in BIG))
def test_map_matches_reference_build(name, oracle_mod):
    O = oracle_mod
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    ov, dims = ALL_MAPS[name]
    ctx = cases.make_ctx(ov)
    o, r = O.Oracle(ctx), O.Ref(ctx)
    assert o.generateMapForPlane(*dims, 0) and r.generateMapForPlane(*dims, 0)
    assert np.array_equal(o.map(0).view(np.uint32), r.map(0).view(np.uint32))


@pytest.mark.parametrize("name", sorted(cases.LOWPASS_CASES))
def test_lowpass_config(name, oracle_mod, golden):
    O = oracle_mod
    ov, dims = cases.LOWPASS_CASES[name]
    ctx = cases.make_ctx(ov)
    o = O.Oracle(ctx)
    assert o.generateMapForPlane(*dims, 0)
    segs = o.segments(0)
    g = golden["lowpass"][name]
    assert len(segs) == g["count"]
    rects = np.array([s[:4] for s in segs], np.int32).reshape(-1, 4)
    kbits = np.concatenate([np.concatenate([s[4], s[5]]) for s in segs]).astype(np.float32)
    assert hx(O.fnv1a64(rects)) == g["rects"]
    assert hx(O.fnv1a64(kbits)) == g["kernels"]
    if O.ref_available():
        r = O.Ref(ctx)
        assert r.generateMapForPlane(*dims, 0)
        rs = r.segments(0)
        assert len(rs) == len(segs)
        for a, b in zip(segs, rs):
            assert a[:4] == b[:4]
            assert np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32))
            assert np.array_equal(a[5].view(np.uint32), b[5].view(np.uint32))


def test_cfg3_integer_kernels_match_survey_appendix_b(oracle_mod):
    """480 segments, kY = [68,120,68], kX per band."""
    O = oracle_mod
    ov, dims = cases.LOWPASS_CASES["cfg3_luma"]
    o = O.Oracle(cases.make_ctx(ov))
    assert o.generateMapForPlane(*dims, 0)
    segs = o.segments(0)
    assert len(segs) == 480
    assert [s[1] for s in segs[::32]] == [896, 768, 640, 512, 384, 256, 128, 0, 1024, 1152, 1280, 1408, 1536, 1664, 1792]
    expect = {896: [120, 68], 768: [118, 69], 640: [101, 63, 15], 512: [90, 62, 20], 384: [78, 61, 28],
              256: [58, 50, 33, 16], 128: [35, 33, 28, 21, 15, 9, 5],
              0: [12, 12, 11, 11, 11, 10, 9, 9, 8, 7, 6, 6, 5, 4, 4, 3, 2, 2, 2],
              1024: [119, 69], 1152: [101, 63, 15], 1280: [91, 62, 20], 1408: [78, 61, 28],
              1536: [58, 50, 33, 16], 1664: [35, 33, 28, 21, 15, 9, 5],
              1792: [12, 12, 12, 11, 11, 10, 10, 9, 8, 7, 6, 6, 5, 4, 4, 3, 2, 2]}
    for s in segs:
        kx = np.rint(s[4].astype(np.float64) * 256).astype(int)
        ky = np.rint(s[5].astype(np.float64) * 256).astype(int)
        assert list(ky) == [68, 120, 68]
        assert list(kx[len(kx) // 2:]) == expect[s[1]]
        assert s[2:4] == (120, 128)
        assert O.lib().t360o_kernel_type(s[4].ctypes.data_as(O._f32p), len(s[4])) == 5  # SMOOTH|SYMMETRICAL
    assert [float(v).hex() for v in segs[0][5]] == ['0x1.0fe2760000000p-2', '0x1.e03b120000000p-2', '0x1.0fe2760000000p-2']


@pytest.mark.parametrize("name", sorted(ALL_FRAMES))
def test_frame_path_matches_shim_generated_golden(name, oracle_mod, golden):
    O = oracle_mod
    ov, dims, pin, pout = ALL_FRAMES[name]
    in_w, in_h, out_w, out_h = dims
    ctx = cases.make_ctx(ov)
    src = cases.case_input(name, in_w, in_h, pin)
    g = golden["frames"][name]
    assert hx(O.fnv1a64(np.ascontiguousarray(src))) == g["in"]
    for threads in (1, 4):
        o = O.Oracle(ctx, threads=threads)
        assert o.generateMapForPlane(*dims, 0)
        full = np.full((out_h, out_w + pout), 0xA5, np.uint8)
        dst = full[:, :out_w]
        assert o.transformFramePlane(src, dst, 0)
        assert hx(O.fnv1a64(np.ascontiguousarray(dst))) == g["out"]
        assert (full[:, out_w:] == 0xA5).all()   # padding bytes are never written
        if "blurred" in g:
            assert hx(O.fnv1a64(o.filterPlane(src, 0))) == g["blurred"]


def test_interpolation_tables(oracle_mod):
    """Properties of OpenCV's Q15 tables."""
    O = oracle_mod
    for interp, k in ((LINEAR, 2), (CUBIC, 4), (LANCZOS4, 8)):
        t = O.inter_tab(interp).astype(np.int64)
        assert t.shape == (1024, k * k)
        assert (t.sum(axis=1) == 32768).all()          # every entry sum-corrected to exactly 1.0
    cubic = O.inter_tab(CUBIC)
    # frac = 0: centre tap saturates at 32767 and the residue lands on tap (2,2)
    assert cubic[0, 1 * 4 + 1] == 32767 and cubic[0, 2 * 4 + 2] == 1
    lin = O.inter_tab(LINEAR)
    fy, fx = 7, 19
    assert list(lin[fy * 32 + fx]) == [(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32]


def test_border_interpolate(oracle_mod):
    L = oracle_mod.lib()
    WRAP, REPL, R101 = 3, 1, 4
    assert [L.t360o_border_interpolate(p, 10, WRAP) for p in (-11, -10, -1, 0, 9, 10, 25)] == [9, 0, 9, 0, 9, 0, 5]
    assert [L.t360o_border_interpolate(p, 10, REPL) for p in (-3, 4, 12)] == [0, 4, 9]
    assert [L.t360o_border_interpolate(p, 10, R101) for p in (-2, -1, 10, 11)] == [2, 1, 8, 7]


def test_nearest_is_a_pure_gather(oracle_mod):
    """Nearest output = input sampled at round-half-even(map) with wrap on both axes."""
    O = oracle_mod
    ctx = filter_defaults(interpolation_alg=NEAREST, enable_low_pass_filter=0)
    dims = (256, 128, 96, 64)
    o = O.Oracle(ctx)
    assert o.generateMapForPlane(*dims, 0)
    m = o.map(0)
    src = cases.case_input("nn", 256, 128, 0)
    dst = np.zeros((64, 96), np.uint8)
    assert o.transformFramePlane(src, dst, 0)
    ix = np.rint(m[..., 0]).astype(int) % 256
    iy = np.rint(m[..., 1]).astype(int) % 128
    assert np.array_equal(dst, src[iy, ix])


def test_constant_plane_is_preserved(oracle_mod):
    """Weights sum to exactly 32768, so a flat plane stays flat through every kernel."""
    O = oracle_mod
    for interp in (NEAREST, LINEAR, CUBIC, LANCZOS4):
        ctx = filter_defaults(interpolation_alg=interp, enable_low_pass_filter=0)
        o = O.Oracle(ctx)
        assert o.generateMapForPlane(128, 64, 48, 32, 0)
        src = np.full((64, 128), 201, np.uint8)
        dst = np.zeros((32, 48), np.uint8)
        assert o.transformFramePlane(src, dst, 0)
        assert (dst == 201).all()


def test_unknown_interpolation_writes_nothing(oracle_mod):
    """interpolation_alg = 3 hits the reference's default: branch (:780-783): returns true."""
    O = oracle_mod
    ctx = filter_defaults(interpolation_alg=3, enable_low_pass_filter=0)
    o = O.Oracle(ctx)
    assert o.generateMapForPlane(128, 64, 48, 32, 0)
    dst = np.full((32, 48), 9, np.uint8)
    assert o.transformFramePlane(np.zeros((64, 128), np.uint8), dst, 0)
    assert (dst == 9).all()


def test_alpha_plane_quirk_oracle_equals_reference(oracle_mod):
    """4th plane of a yuva420p frame: map index 0 with chroma dimensions (vf_transform360.c:368-397)."""
    O = oracle_mod
    if not O.ref_available():
        pytest.skip("reference build (oracle/_ref) not available here")
    for ov in (dict(enable_low_pass_filter=0), dict(num_vertical_segments=5, num_horizontal_segments=4)):
        ctx = filter_defaults(**ov)
        o, r = O.Oracle(ctx, threads=2), O.Ref(ctx)
        src = np.random.default_rng(7).integers(0, 256, (120, 240), dtype=np.uint8)
        a, b = np.full((64, 96), 0x5A, np.uint8), np.full((64, 96), 0x5A, np.uint8)
        assert o.generateMapForPlane(480, 240, 192, 128, 0) and r.generateMapForPlane(480, 240, 192, 128, 0)
        assert o.transformFramePlane(src, a, 0, 3) and r.transformFramePlane(src, b, 0, 3)
        assert np.array_equal(a, b)


def test_regenerating_a_map_with_other_dims_reference_appends_oracle_replaces(oracle_mod):
    """a second generateMapForPlane on the same index REPLACES the warp map but APPENDS the low-pass
    segments and kernels in the reference (VideoFrameTransform.cpp:237, :290-294, :556); the oracle and the library
    replace them.  The old segments run first and the new ones -- which tile the whole plane -- overwrite whatever the
    old ones wrote, so the planes are identical (old rectangles outside the new plane only print the reference's
    'Could not filter segment' message).  Pinned here against the reference build, both growing and shrinking, with
    enable_multi_threading = 0: with threads the reference runs old and new segments CONCURRENTLY on the same pixels
    (one std::thread per segment, :592-604) and its own output is a race -- replacing is what it computes when the new
    segments win."""
    O = oracle_mod
    if not O.ref_available():
        pytest.skip("/root/reference is not mounted here (the GPU box): the reference build cannot be made")
    ov = dict(num_vertical_segments=5, num_horizontal_segments=4, enable_multi_threading=0)
    for first, second in (((512, 256, 192, 128), (1024, 512, 384, 256)), ((1024, 512, 384, 256), (512, 256, 192, 128))):
        r, o = O.Ref(cases.make_ctx(ov)), O.Oracle(cases.make_ctx(ov))
        for dims in (first, second):
            assert r.generateMapForPlane(*dims, 0) and o.generateMapForPlane(*dims, 0)
        assert len(r.segments(0)) == 2 * len(o.segments(0)) == 40  # appended there, replaced here
        in_w, in_h, out_w, out_h = second
        src = cases.case_input("regen", in_w, in_h, 0)
        a = np.full((out_h, out_w), 0xA5, np.uint8)
        b = np.full((out_h, out_w), 0xA5, np.uint8)
        assert r.transformFramePlane(src, a, 0) and o.transformFramePlane(src, b, 0)
        assert np.array_equal(a, b)
        assert np.array_equal(r.filterPlane(src, 0), o.filterPlane(src, 0))
        r.close()
        o.close()
