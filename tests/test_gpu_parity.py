"""GPU parity suite (-m gpu): the HIP path, called through the exported C ABI, against the CPU
oracle on the same inputs and against the committed golden vectors.

Bars (BASELINE.json north_star): warp maps bit-exact (float bits), nearest bit-exact,
bilinear / bicubic / Lanczos4 / low-pass within +-1 LSB of the reference path -- the HIP kernels
use the oracle's integer formulation, so the tests below demand BIT-EXACT equality with the
oracle everywhere; the +-1 LSB is left to the oracle-vs-real-OpenCV uncertainty
(oracle/t360_oracle_cv.c header).
"""
import os

import numpy as np
import pytest

from tests import cases
from transform360_amd.abi import CUBIC, LANCZOS4, LINEAR, NEAREST, chroma_dims, filter_defaults

pytestmark = pytest.mark.gpu


def hx(v):
    return "%016x" % v


@pytest.fixture(scope="module")
def T(gpu_lib):
    import torch

    from transform360_amd import handler
    assert torch.cuda.is_available()
    return handler


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ready():
    """Tensors are created on torch's stream, the handle works on its own non-blocking stream: the
    reference ABI takes buffers that are READY (include/Transform360/VideoFrameTransformHandler.h)."""
    import torch
    torch.cuda.synchronize()


def padded_cuda(h, w, pad, fill):
    import torch
    full = torch.full((h, w + pad), fill, dtype=torch.uint8, device="cuda")
    return full, full[:, :w]


# ---------------------------------------------------------------- projection kernel
ALL_MAPS = {**cases.MAP_CASES, **cases.LAYOUT_MAP_CASES}
ALL_FRAMES = {**cases.FRAME_CASES, **cases.LAYOUT_FRAME_CASES, **cases.SUPERSAMPLE_FRAME_CASES}


@pytest.mark.parametrize("name", sorted(ALL_MAPS))
def test_map_bit_exact(name, T, oracle_mod, golden):
    O = oracle_mod
    ov, dims = ALL_MAPS[name]
    ctx = cases.make_ctx(ov)
    with T.VideoFrameTransform(ctx) as t:
        assert t.generateMapForPlane(*dims, 0)
        m = t.map(0)
    g = golden["maps"][name]
    assert m.shape == (g["h"], g["w"], 2)
    q, nn = O.quantize_map(m)
    got = (hx(O.fnv1a64(m)), hx(O.fnv1a64(q)), hx(O.fnv1a64(nn)))
    if got != (g["f32"], g["q"], g["nn"]):
        o = O.Oracle(ctx)
        assert o.generateMapForPlane(*dims, 0)
        ref = o.map(0)
        bad = np.argwhere(m.view(np.uint32) != ref.view(np.uint32))
        r, c, k = bad[0]
        pytest.fail("%s: %d of %d coordinates differ from the reference map; first at (%d,%d)[%d]: "
                    "gpu %s ref %s" % (name, len(bad), m.size, r, c, k, float(m[r, c, k]).hex(), float(ref[r, c, k]).hex()))


def test_unsupported_request_is_refused_not_faked(T):
    from transform360_amd.abi import LAYOUT_N
    with T.VideoFrameTransform(filter_defaults(output_layout=LAYOUT_N)) as t:
        assert not t.generateMapForPlane(1024, 512, 384, 256, 0)
    # scale factors that overflow or are not numbers, and maps beyond 2^28 entries, are refused before anything is allocated
    for ov in (dict(width_scale_factor=float("nan")), dict(height_scale_factor=float("inf")), dict(width_scale_factor=1e9),
               dict(width_scale_factor=-1.0), dict(height_scale_factor=0.0)):
        with T.VideoFrameTransform(filter_defaults(enable_low_pass_filter=0, **ov)) as t:
            assert not t.generateMapForPlane(1024, 512, 384, 256, 0)
    with T.VideoFrameTransform(filter_defaults(enable_low_pass_filter=0)) as t:
        assert not t.generateMapForPlane(1024, 512, 20000, 20000, 0)
        assert not t.generateMapForPlane(1024, 512, 384, 256, -1)
        assert not t.generateMapForPlane(1024, 512, 384, 256, 1000)
        assert not t.generateMapForPlane(0, 512, 384, 256, 0)
        assert not t.generateMapForPlane(1024, 512, 384, -3, 0)


def test_output_size_given_at_call_time_is_resized_like_the_reference(T, oracle_mod):
    # transformPlane resizes the warped image to WHATEVER output size the call names (VideoFrameTransform.cpp:735-737,
    # 759-776), not only to the size generateMapForPlane was told: here a 192x128 map (factors 0.5) lands in a
    # 200x100 plane -- INTER_AREA enlarging in x, shrinking in y
    O = oracle_mod
    ctx = filter_defaults(width_scale_factor=0.5, height_scale_factor=0.5, enable_low_pass_filter=0)
    o = O.Oracle(ctx, threads=2)
    src = np.random.default_rng(11).integers(0, 256, (512, 1024), dtype=np.uint8)
    want = np.zeros((100, 200), np.uint8)
    assert o.generateMapForPlane(1024, 512, 384, 256, 0) and o.transformFramePlane(src, want, 0, 0)
    with T.VideoFrameTransform(ctx) as t:
        assert t.generateMapForPlane(1024, 512, 384, 256, 0)
        dsrc, ddst = dev(src), dev(np.zeros((100, 200), np.uint8))
        _ready()
        assert t.transformFramePlane(dsrc, ddst, 0) and t.synchronize()
        assert np.array_equal(ddst.cpu().numpy(), want)
        # and back to the size the map was generated for (the tables follow the call)
        want2 = np.zeros((256, 384), np.uint8)
        assert o.transformFramePlane(src, want2, 0, 0)
        ddst2 = dev(np.zeros((256, 384), np.uint8))
        assert t.transformFramePlane(dsrc, ddst2, 0) and t.synchronize()
        assert np.array_equal(ddst2.cpu().numpy(), want2)


# ---------------------------------------------------------------- low-pass configuration
@pytest.mark.parametrize("name", sorted(cases.LOWPASS_CASES))
def test_lowpass_config_matches(name, T, oracle_mod, golden):
    O = oracle_mod
    ov, dims = cases.LOWPASS_CASES[name]
    with T.VideoFrameTransform(cases.make_ctx(ov)) as t:
        assert t.generateMapForPlane(*dims, 0)
        segs = t.segments(0)
    g = golden["lowpass"][name]
    assert len(segs) == g["count"]
    rects = np.array([s[:4] for s in segs], np.int32).reshape(-1, 4)
    kbits = np.concatenate([np.concatenate([s[4], s[5]]) for s in segs]).astype(np.float32)
    assert hx(O.fnv1a64(rects)) == g["rects"]
    assert hx(O.fnv1a64(kbits)) == g["kernels"]


# ---------------------------------------------------------------- whole plane, device pointers
@pytest.mark.parametrize("name", sorted(ALL_FRAMES))
def test_frame_case_device_pointers(name, T, oracle_mod, golden):
    O = oracle_mod
    ov, dims, pin, pout = ALL_FRAMES[name]
    in_w, in_h, out_w, out_h = dims
    ctx = cases.make_ctx(ov)
    src = cases.case_input(name, in_w, in_h, pin)
    # oracle
    o = O.Oracle(ctx, threads=4)
    assert o.generateMapForPlane(*dims, 0)
    # BARREL outputs are remapped with BORDER_TRANSPARENT: pixels without a mapping keep what the
    # destination held, so every side starts from the same 0xA5 fill (as the golden vectors did)
    want = np.full((out_h, out_w), 0xA5, np.uint8)
    assert o.transformFramePlane(src, want, 0)
    assert hx(O.fnv1a64(want)) == golden["frames"][name]["out"]
    # HIP path through the reference ABI with device pointers and the same padded strides
    dsrc_full = dev(np.ascontiguousarray(src.base if src.base is not None else src).reshape(in_h, in_w + pin))
    dsrc = dsrc_full[:, :in_w]
    dfull, ddst = padded_cuda(out_h, out_w, pout, 0xA5)
    with T.VideoFrameTransform(ctx) as t:
        assert t.generateMapForPlane(*dims, 0)
        if ctx.enable_low_pass_filter:
            import torch
            blur = torch.zeros((in_h, in_w), dtype=torch.uint8, device="cuda")
            _ready()
            assert t.filterPlane(dsrc, blur, 0) and t.synchronize()
            wantb = o.filterPlane(src, 0)
            gotb = blur.cpu().numpy()
            assert np.array_equal(gotb, wantb), "low-pass stage: %d px differ, max |d| %d" % (
                (gotb != wantb).sum(), np.abs(gotb.astype(int) - wantb.astype(int)).max())
        _ready()
        assert t.transformFramePlane(dsrc, ddst, 0, 0)
    got = ddst.cpu().numpy()
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() == 0, "%s: %d px differ, max |d| = %d" % (name, (diff > 0).sum(), diff.max())
    assert (dfull[:, out_w:] == 0xA5).all().item()   # only `width` bytes per row may be written


# ---------------------------------------------------------------- host pointers (the literal ABI)
@pytest.mark.parametrize("name", ["cubic", "nearest", "cubic_lpf_32x15", "tiny"])
def test_frame_case_host_pointers(name, T, golden, oracle_mod):
    ov, dims, pin, pout = cases.FRAME_CASES[name]
    in_w, in_h, out_w, out_h = dims
    src = cases.case_input(name, in_w, in_h, pin)
    full = np.full((out_h, out_w + pout), 0xA5, np.uint8)
    dst = full[:, :out_w]
    with T.VideoFrameTransform(cases.make_ctx(ov)) as t:
        assert t.generateMapForPlane(*dims, 0)
        _ready()
        assert t.transformFramePlane(src, dst, 0, 0)     # numpy arrays = host pointers
    assert hx(oracle_mod.fnv1a64(np.ascontiguousarray(dst))) == golden["frames"][name]["out"]
    assert (full[:, out_w:] == 0xA5).all()


# ---------------------------------------------------------------- vf_transform360.c call sequence
def test_filter_call_sequence_yuv420p(T, oracle_mod):
    """_new, two _generateMapForPlane calls (luma dims, chroma dims via ceil-shift), then three
    _transformFramePlane calls per frame with padded linesizes (vf_transform360.c:141-162, 368-397)."""
    O = oracle_mod
    in_w, in_h, out_w, out_h = 1280, 640, 768, 512
    ctx = filter_defaults(num_vertical_segments=15, num_horizontal_segments=32)
    cw, ch = chroma_dims(in_w, in_h)
    ocw, och = chroma_dims(out_w, out_h)
    o = O.Oracle(ctx, threads=4)
    with T.VideoFrameTransform(ctx) as t:
        for idx, d in enumerate([(in_w, in_h, out_w, out_h), (cw, ch, ocw, och)]):
            assert t.generateMapForPlane(*d, idx) and o.generateMapForPlane(*d, idx)
        for frame in range(2):
            for plane in range(3):
                idx = 1 if plane in (1, 2) else 0
                iw, ih, ow, oh = (in_w, in_h, out_w, out_h) if plane == 0 else (cw, ch, ocw, och)
                src = cases.case_input("seq%d_%d" % (frame, plane), iw, ih, 48)
                want = np.zeros((oh, ow), np.uint8)
                assert o.transformFramePlane(src, want, idx, plane)
                full = np.full((oh, ow + 32), 0x11, np.uint8)
                dst = full[:, :ow]
                _ready()
                assert t.transformFramePlane(src, dst, idx, plane)
                assert np.array_equal(dst, want)
                assert (full[:, ow:] == 0x11).all()


# ---------------------------------------------------------------- batch entry point
def _batch_case(T, O, ov, n=5, dims=(960, 480, 384, 256), extra_pad=64, threads=4, pipelined=False, fused=False):
    """n frames x 3 planes through T360_transformFrames == per-plane oracle calls.  pipelined: the same batch three more
    times through T360_transformFramesPipelined (three lanes, three buffers) == the plain call's output, bit for bit."""
    import torch
    in_w, in_h, out_w, out_h = dims
    ctx = filter_defaults(**ov)
    lin = T.FrameLayout(in_w, in_h, extra_pad=extra_pad)
    lout = T.FrameLayout(out_w, out_h)
    d_in = torch.empty(n * lin.frame_bytes, dtype=torch.uint8, device="cuda")
    for k in range(n):
        T.fill_noise(d_in[k * lin.frame_bytes:(k + 1) * lin.frame_bytes], T.frame_seed(k))
    d_out = torch.full((n * lout.frame_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    o = O.Oracle(ctx, threads=threads)
    with T.VideoFrameTransform(ctx) as t:
        if fused:
            assert t.setFusedLowpass(True)   # before the maps: tiles that can filter their own footprint in LDS do
        for idx, k in ((0, 0), (1, 1)):
            d = (*lin.dims[k], *lout.dims[k])
            assert t.generateMapForPlane(*d, idx) and o.generateMapForPlane(*d, idx)
        assert t.setStream(torch.cuda.current_stream())
        assert t.transformFrames(d_in, lin.frame_bytes, d_out, lout.frame_bytes, n, t.plane_descs(lin, lout))
        assert t.synchronize()
        if fused is True:
            assert "remap_fused_kernel" in t.lastKernel(), t.lastKernel()   # the path under test ran
        ran_fused = "remap_fused_kernel" in t.lastKernel()
        if ov.get("num_horizontal_segments") == 32 and dims[0] == 3840 and extra_pad == 0:
            # BASELINE config 3 at full size: the Y, U and V low-pass of the batch is ONE launch (no test pinned it)
            assert t.lastLowpassPath() == "merged", t.lastLowpassPath()
        if pipelined:
            assert t.setPipelineDepth(3)
            outs = [torch.full((n * lout.frame_bytes,), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(3)]
            torch.cuda.synchronize()
            for o3 in outs:
                assert t.transformFramesPipelined(d_in, lin.frame_bytes, o3, lout.frame_bytes, n, t.plane_descs(lin, lout))
            assert t.synchronize()
            for o3 in outs:
                assert torch.equal(o3, d_out), "a pipelined call differs from the plain one"
        h_in = d_in.cpu().numpy()
        h_out = d_out.cpu().numpy()
        for k in range(n):
            fin = h_in[k * lin.frame_bytes:(k + 1) * lin.frame_bytes]
            # the device generator and its host restatement agree byte for byte
            assert np.array_equal(fin, T.noise_bytes(lin.frame_bytes, T.frame_seed(k)))
            fout = h_out[k * lout.frame_bytes:(k + 1) * lout.frame_bytes]
            for p in range(3):
                # same initial content as d_out: BARREL outputs (BORDER_TRANSPARENT) leave unmapped pixels alone
                want = np.full((lout.dims[p][1], lout.dims[p][0]), 0x5A, np.uint8)
                assert o.transformFramePlane(lin.plane_view(fin, p), want, 1 if p else 0, p)
                assert np.array_equal(lout.plane_view(fout, p), want), (k, p)
    return ran_fused


def test_batch_equals_per_plane_calls(T, oracle_mod):
    for ov in (dict(enable_low_pass_filter=0), dict(num_vertical_segments=5, num_horizontal_segments=4)):
        _batch_case(T, oracle_mod, ov)


@pytest.mark.parametrize("ov,depth", [(dict(enable_low_pass_filter=0), 2),
                                      (dict(num_vertical_segments=5, num_horizontal_segments=4), 3),
                                      (dict(enable_low_pass_filter=0, width_scale_factor=2.0, height_scale_factor=2.0), 4)])
def test_pipelined_calls_equal_plain_calls_and_the_oracle(ov, depth, T, oracle_mod):
    """T360_transformFramesPipelined: a stream of independent batches issued round-robin on the handle's internal streams
    (each lane has its own low-pass / supersample scratch planes).  Seven batches of different frames, output buffers
    reused every `depth` calls as the header allows: every batch equals the plain T360_transformFrames result bit for
    bit, and one batch is compared with the oracle directly; pipelineJoin() orders a consumer on the handle's stream."""
    import torch
    O = oracle_mod
    in_w, in_h, out_w, out_h = 960, 480, 384, 256
    nb, n = 7, 3
    ctx = filter_defaults(**ov)
    lin, lout = T.FrameLayout(in_w, in_h), T.FrameLayout(out_w, out_h)
    batches = []
    for b in range(nb):
        d = torch.empty(n * lin.frame_bytes, dtype=torch.uint8, device="cuda")
        for k in range(n):
            T.fill_noise(d[k * lin.frame_bytes:(k + 1) * lin.frame_bytes], T.frame_seed(100 * b + k))
        batches.append(d)
    torch.cuda.synchronize()
    with T.VideoFrameTransform(ctx) as t:
        for idx, k in ((0, 0), (1, 1)):
            assert t.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
        descs = t.plane_descs(lin, lout)
        assert t.setStream(torch.cuda.current_stream())
        want = []
        for b in range(nb):
            o_ = torch.zeros(n * lout.frame_bytes, dtype=torch.uint8, device="cuda")
            assert t.transformFrames(batches[b], lin.frame_bytes, o_, lout.frame_bytes, n, descs)
            want.append(o_)
        assert t.synchronize()
        assert not t.setPipelineDepth(0) and not t.setPipelineDepth(5)
        assert t.setPipelineDepth(depth)
        outs = [torch.zeros(n * lout.frame_bytes, dtype=torch.uint8, device="cuda") for _ in range(depth)]
        got = []
        for b in range(nb):
            buf = outs[b % depth]
            if b >= depth:
                # the buffer's previous content must be saved before its lane overwrites it: order the copy after the
                # pipelined calls issued so far
                assert t.pipelineJoin()
                got.append(buf.clone())
            assert t.transformFramesPipelined(batches[b], lin.frame_bytes, buf, lout.frame_bytes, n, descs)
        assert t.pipelineJoin()
        for b in range(nb - depth, nb):
            got.append(outs[b % depth].clone())
        assert t.synchronize()
        torch.cuda.synchronize()
        for b in range(nb):
            assert torch.equal(got[b], want[b]), "pipelined batch %d differs from the plain call" % b
        # the same seven calls issued by the library's own loop (T360_transformFramesPipelinedMany), one buffer per call
        many = [torch.zeros(n * lout.frame_bytes, dtype=torch.uint8, device="cuda") for _ in range(nb)]
        torch.cuda.synchronize()
        assert t.transformFramesPipelinedMany(batches, lin.frame_bytes, many, lout.frame_bytes, n, descs)
        assert t.synchronize()
        for b in range(nb):
            assert torch.equal(many[b], want[b]), "batch %d of the bulk call differs from the plain call" % b
    o = O.Oracle(ctx, threads=4)
    for idx, k in ((0, 0), (1, 1)):
        assert o.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
    h_in, h_out = batches[nb - 1].cpu().numpy(), got[nb - 1].cpu().numpy()
    for k in range(n):
        fin = h_in[k * lin.frame_bytes:(k + 1) * lin.frame_bytes]
        fout = h_out[k * lout.frame_bytes:(k + 1) * lout.frame_bytes]
        for p in range(3):
            ref = np.zeros((lout.dims[p][1], lout.dims[p][0]), np.uint8)
            assert o.transformFramePlane(lin.plane_view(fin, p), ref, 1 if p else 0, p)
            assert np.array_equal(lout.plane_view(fout, p), ref), (k, p)
    o.close()


def test_pipelined_calls_that_name_a_map_with_two_plane_sizes(T):
    """Consecutive pipelined calls that name map 0 with two plane sizes (the filter's alpha-plane quirk as a stream): the
    per-map low-pass tile lists and INTER_AREA tables are rebuilt in place for every call, so the library must not start
    rewriting them while the previous call still runs on another lane.  Every output equals the plain call's."""
    import torch
    from transform360_amd import _lib
    ctx = filter_defaults(num_vertical_segments=5, num_horizontal_segments=4)
    n = 4
    lin, lout = T.FrameLayout(480, 240, planes=1), T.FrameLayout(192, 128, planes=1)
    lin2, lout2 = T.FrameLayout(240, 120, planes=1), T.FrameLayout(96, 64, planes=1)

    def desc(a, b):
        d = (_lib.T360PlaneDesc * 1)()
        d[0] = _lib.T360PlaneDesc(in_offset=0, out_offset=0, in_stride=a.strides[0], out_stride=b.strides[0], in_width=a.dims[0][0],
                                  in_height=a.dims[0][1], out_width=b.dims[0][0], out_height=b.dims[0][1], map_index=0)
        return d
    shapes = [(lin, lout, desc(lin, lout)), (lin2, lout2, desc(lin2, lout2))]
    with T.VideoFrameTransform(ctx) as t:
        assert t.generateMapForPlane(480, 240, 192, 128, 0)
        ins, want = [], []
        for b in range(6):
            a, o_, d = shapes[b & 1]
            x = torch.empty(n * a.frame_bytes, dtype=torch.uint8, device="cuda")
            for k in range(n):
                T.fill_noise(x[k * a.frame_bytes:(k + 1) * a.frame_bytes], T.frame_seed(900 + 10 * b + k))
            y = torch.zeros(n * o_.frame_bytes, dtype=torch.uint8, device="cuda")
            _ready()
            assert t.transformFrames(x, a.frame_bytes, y, o_.frame_bytes, n, d) and t.synchronize()
            ins.append(x)
            want.append(y)
        assert t.setPipelineDepth(3)
        got = [torch.zeros_like(w) for w in want]
        _ready()
        for b in range(6):
            a, o_, d = shapes[b & 1]
            assert t.transformFramesPipelined(ins[b], a.frame_bytes, got[b], o_.frame_bytes, n, d)
        assert t.synchronize()
        for b in range(6):
            assert torch.equal(got[b], want[b]), "pipelined call %d differs" % b


@pytest.mark.parametrize("interp", [NEAREST, LINEAR, LANCZOS4])
def test_batch_other_interpolations_tiled(interp, T, oracle_mod):
    # the LDS-tiled DMA-ring kernel instantiated for 1-, 2- and 8-tap stencils (frames 16-byte friendly)
    _batch_case(T, oracle_mod, dict(enable_low_pass_filter=0, interpolation_alg=interp), n=3, extra_pad=0)
    # and the general gather for buffers that are not (odd padding)
    _batch_case(T, oracle_mod, dict(enable_low_pass_filter=0, interpolation_alg=interp), n=2, extra_pad=40)


def test_alpha_plane_quirk_matches(T, oracle_mod):
    """The filter hands a 4th (alpha) plane to map index 0 with CHROMA dimensions (vf_transform360.c:368-397): the luma
    map then samples a plane a quarter of its size (BORDER_WRAP), the output takes the resize branch, and with the
    low-pass on every segment that does not fit the smaller plane is skipped with a message.  Same bytes as the oracle
    (which agrees with the reference build on this case, tests/test_oracle.py)."""
    import torch
    O = oracle_mod
    for ov in (dict(enable_low_pass_filter=0), dict(num_vertical_segments=5, num_horizontal_segments=4)):
        ctx = filter_defaults(**ov)
        o = O.Oracle(ctx, threads=2)
        src = np.random.default_rng(7).integers(0, 256, (120, 240), dtype=np.uint8)
        want = np.full((64, 96), 0x5A, np.uint8)
        assert o.generateMapForPlane(480, 240, 192, 128, 0) and o.transformFramePlane(src, want, 0, 3)
        with T.VideoFrameTransform(ctx) as t:
            assert t.generateMapForPlane(480, 240, 192, 128, 0)
            dsrc = torch.from_numpy(src).cuda()
            ddst = torch.full((64, 96), 0x5A, dtype=torch.uint8, device="cuda")
            _ready()
            assert t.transformFramePlane(dsrc, ddst, 0, 3) and t.synchronize()
            assert np.array_equal(ddst.cpu().numpy(), want)


@pytest.mark.parametrize("ov", [dict(enable_low_pass_filter=0), dict(num_vertical_segments=5, num_horizontal_segments=4),
                                dict(enable_low_pass_filter=0, width_scale_factor=1.5, height_scale_factor=2.0)])
def test_one_map_with_two_plane_sizes_in_one_batch(ov, T, oracle_mod):
    """yuva420p through the batch entry point: Y and A both name map 0, A with chroma dimensions (the filter's alpha-plane
    quirk, vf_transform360.c:368-397).  The per-map tables that depend on the plane size (INTER_AREA tables, low-pass tile
    lists) exist once per map: the library must not let the second size overwrite them before the first plane's
    kernels have run.  Every plane of every frame equals per-plane oracle calls."""
    import torch
    from transform360_amd import _lib
    O = oracle_mod
    ctx = filter_defaults(**ov)
    n = 3
    in_w, in_h, out_w, out_h = 480, 240, 192, 128
    lin = T.FrameLayout(in_w, in_h, planes=4)
    lout = T.FrameLayout(out_w, out_h, planes=4)
    o = O.Oracle(ctx, threads=4)
    with T.VideoFrameTransform(ctx) as t:
        for idx, k in ((0, 0), (1, 1)):
            assert t.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx) and o.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
        frames = [T.noise_bytes(lin.frame_bytes, T.frame_seed(70 + j)) for j in range(n)]
        d_in = torch.from_numpy(np.concatenate(frames)).cuda()
        d_out = torch.full((n * lout.frame_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
        descs = (_lib.T360PlaneDesc * 4)()
        for k in range(4):
            descs[k] = _lib.T360PlaneDesc(in_offset=lin.offsets[k], out_offset=lout.offsets[k], in_stride=lin.strides[k],
                                          out_stride=lout.strides[k], in_width=lin.dims[k][0], in_height=lin.dims[k][1],
                                          out_width=lout.dims[k][0], out_height=lout.dims[k][1], map_index=1 if k in (1, 2) else 0)
        _ready()
        assert t.transformFrames(d_in, lin.frame_bytes, d_out, lout.frame_bytes, n, descs) and t.synchronize()
        got = d_out.cpu().numpy()
    for j in range(n):
        for k in range(4):
            want = np.full((lout.dims[k][1], lout.dims[k][0]), 0x5A, np.uint8)
            assert o.transformFramePlane(lin.plane_view(frames[j], k), want, 1 if k in (1, 2) else 0, k)
            have = lout.plane_view(got[j * lout.frame_bytes:(j + 1) * lout.frame_bytes], k)
            assert np.array_equal(have, want), "frame %d plane %d" % (j, k)
    o.close()


def test_gather_plans_are_built_by_the_first_call_that_needs_them(T):
    """generateMapForPlane keeps a host copy of the LUT; the plan for short batches is built by the first short call, the
    plan for long batches by the first long one (T360_getPlanStats reports whichever exists, the long one first)."""
    import torch
    ctx = filter_defaults(enable_low_pass_filter=0)
    with T.VideoFrameTransform(ctx) as t:
        assert t.generateMapForPlane(960, 480, 384, 256, 0)
        assert t.planStats(0) is None
        src = torch.zeros((480, 960), dtype=torch.uint8, device="cuda")
        dst = torch.zeros((256, 384), dtype=torch.uint8, device="cuda")
        _ready()
        assert t.transformFramePlane(src, dst, 0)
        small = t.planStats(0)
        assert small is not None and t.lastKernel() == "remap_tiled_kernel<4, 38, 4>"
        lin, lout = T.FrameLayout(960, 480, planes=1), T.FrameLayout(384, 256, planes=1)
        n = 32
        d_in = torch.zeros(n * lin.frame_bytes, dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(n * lout.frame_bytes, dtype=torch.uint8, device="cuda")
        _ready()
        assert t.transformFrames(d_in, lin.frame_bytes, d_out, lout.frame_bytes, n, t.plane_descs(lin, lout)) and t.synchronize()
        assert t.lastKernel() == "remap_tiled_kernel<4, 76, 8>"
        big = t.planStats(0)
        assert big is not None and big["staged_tiles"] < small["staged_tiles"]  # 128x16 tiles instead of 64x16


@pytest.mark.parametrize("n", [64, 65, 129])
def test_batch_frame_count_boundaries(n, T, oracle_mod):
    # batches longer than one run of frames per workgroup (64): a second, shorter run; the 16-frame runs of the last tiles
    _batch_case(T, oracle_mod, dict(enable_low_pass_filter=0), n=n, dims=(640, 320, 384, 256), extra_pad=0, threads=8)


def test_host_buffers_that_come_and_go():
    """Host planes are allocated, used a few times, freed, and new ones of other sizes take their addresses
    (tests/soak/host_soak.py, in a child process: a stale pinned range -- what a cache of hipHostRegister'ed caller buffers
    turns into -- ends in a GPU memory fault, which kills the process that triggers it)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "soak", "host_soak.py"), "150"], cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "150 host-pointer calls, 0 wrong" in r.stdout


# The shipped library reads no environment.  The instrumented build (make instr: -DT360_INSTRUMENT) does, for A/B
# runs of ring geometries and plan options; every one of those switches must leave the pixels alone.  It is a second
# library, so it runs in a child process (tests/run_variants.py) with T360_LIB pointing at it.
VARIANTS = [
    {"T360_WAVES": "4", "T360_RING_KB": "31", "T360_MAX_PIECES": "12"}, {"T360_WAVES": "4", "T360_RING_KB": "26", "T360_MAX_PIECES": "12"},
    {"T360_MAX_PIECES": "8"}, {"T360_MAX_PIECES": "4"}, {"T360_WAVES": "4", "T360_RING_KB": "38", "T360_MAX_PIECES": "16"},
    {"T360_ROW_ALIGN": "1"}, {"T360_ROW_ALIGN": "4"}, {"T360_STRIPS": "120"}, {"T360_STRIPS": "1000"}, {"T360_WIDE64": "0"},
    {"T360_WIDE64": "1000"}, {"T360_BAND": "1"}, {"T360_ROW_PAD": "2"}, {"T360_FRAMES_PER_BLOCK": "2"},
    {"T360_FRAMES_PER_BLOCK": "3", "T360_WAVES": "4", "T360_RING_KB": "31", "T360_MAX_PIECES": "8"}, {"T360_NO_TILED": "1"}, {"T360_NO_FAST_LOWPASS": "1"},
    {"T360_NO_WIDE_LOWPASS": "1"}, {"T360_SMALL_BATCH": "1000"},
    # round 4: 256x8 tiles (kTileWide256) wherever they fit / where they touch fewer lines; extra LDS per workgroup
    {"T360_WIDE256": "1000"}, {"T360_WIDE256": "100", "T360_COST_LINES": "1"}, {"T360_LDS_PAD": "8192"},
    # scatter tiles (blocks grouped by source position), strips of 1 / 2 / 4 / 8 lines
    {"T360_SCATTER": "1"}, {"T360_SCATTER": "2"}, {"T360_SCATTER": "4"}, {"T360_SCATTER": "8"},
]


def test_instrumented_build_variants_are_bit_identical():
    import json
    import subprocess
    import sys
    from transform360_amd import _lib
    instr = os.path.join(os.path.dirname(_lib.LIB_PATH), "libTransform360_instr.so")
    assert os.path.exists(instr), "build it with `make -C transform360_amd/csrc instr`"
    env = dict(os.environ, T360_LIB=instr)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "run_variants.py"), json.dumps(VARIANTS)],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert "variants ok: %d" % len(VARIANTS) in r.stdout


def test_shipped_library_reads_no_environment(T, oracle_mod, monkeypatch):
    from transform360_amd import _lib
    assert _lib.load().T360_buildFlags() == 0
    # a switch of the instrumented build must be inert here: same kernel instantiation as without it
    monkeypatch.setenv("T360_RING_KB", "50")
    monkeypatch.setenv("T360_NO_TILED", "1")
    import torch
    with T.VideoFrameTransform(filter_defaults(enable_low_pass_filter=0)) as t:
        assert t.generateMapForPlane(960, 480, 384, 256, 0)
        src = torch.zeros((480, 960), dtype=torch.uint8, device="cuda")
        dst = torch.zeros((256, 384), dtype=torch.uint8, device="cuda")
        _ready()
        assert t.transformFramePlane(src, dst, 0)
        assert t.lastKernel() == "remap_tiled_kernel<4, 38, 4>"  # a single frame: the short-batch plan (64 frames: <4, 76, 8>)


# ---------------------------------------------------------------- full-size configs (BASELINE)
def _full_size_case(T, O, ov, luma_dims, n_check_rows=None):
    import torch
    in_w, in_h, out_w, out_h = luma_dims
    ctx = filter_defaults(**ov)
    o = O.Oracle(ctx, threads=8)
    assert o.generateMapForPlane(*luma_dims, 0)
    src = T.noise_bytes(in_w * in_h, 0xBA5E).reshape(in_h, in_w)
    want = np.zeros((out_h, out_w), np.uint8)
    assert o.transformFramePlane(src, want, 0)
    with T.VideoFrameTransform(ctx) as t:
        assert t.generateMapForPlane(*luma_dims, 0)
        dsrc = dev(src)
        ddst = torch.zeros((out_h, out_w), dtype=torch.uint8, device="cuda")
        _ready()
        assert t.transformFramePlane(dsrc, ddst, 0)
        got = ddst.cpu().numpy()
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() == 0, "%d px differ, max |d| = %d" % ((diff > 0).sum(), diff.max())


def test_config1_nearest_full_size(T, oracle_mod):
    _full_size_case(T, oracle_mod, dict(interpolation_alg=NEAREST, enable_low_pass_filter=0), (1920, 960, 768, 512))


def test_config2_bicubic_full_size(T, oracle_mod):
    _full_size_case(T, oracle_mod, dict(interpolation_alg=CUBIC, enable_low_pass_filter=0), (3840, 1920, 1536, 1024))


def test_config3_bicubic_lowpass_full_size(T, oracle_mod):
    _full_size_case(T, oracle_mod, dict(num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1),
                    (3840, 1920, 1536, 1024))


def test_config4_lanczos_tb_full_size(T, oracle_mod):
    from tests.cases import TB
    _full_size_case(T, oracle_mod, dict(interpolation_alg=LANCZOS4, enable_low_pass_filter=0, **TB),
                    (7680, 3840, 3072, 4096))


# ---------------------------------------------------------------- the benchmarked path itself
# bench.py's step is ONE T360_transformFrames call over a batch of full-size yuv420p frames (Y, U and V fused into
# one launch; 16 frames per workgroup; the 4K tile plan).  Same call here, with batches that cross the
# frames-per-workgroup boundary, every plane of every frame against per-plane oracle calls.
def _threads():
    return max(4, min(32, os.cpu_count() or 4))


# Batches of 24 frames or more run the 8-wave plan (128x16 tiles, what bench.py measures), shorter ones the 4-wave plan
@pytest.mark.parametrize("n", [27, 19])
def test_config2_batch_full_size_all_planes(n, T, oracle_mod):
    _batch_case(T, oracle_mod, dict(interpolation_alg=CUBIC, enable_low_pass_filter=0), n=n,
                dims=(3840, 1920, 1536, 1024), extra_pad=0, threads=_threads())


def test_config3_batch_full_size_all_planes(T, oracle_mod):
    _batch_case(T, oracle_mod, dict(interpolation_alg=CUBIC, num_vertical_segments=15, num_horizontal_segments=32,
                                    adjust_kernel=1, enable_multi_threading=1), n=25,
                dims=(3840, 1920, 1536, 1024), extra_pad=0, threads=_threads())


# T360_setFusedLowpass (round 6): the low-pass of a tile's own footprint done in LDS by the gather workgroup
# (remap_fused_kernel; reference filterPlane feeding remap on one plane, VideoFrameTransform.cpp:727-733 + :748-754,
# :173-204) -- same bytes as the two-pass path and the oracle, on BASELINE config 3 at full size, on the filter's default
# 5 x 1 segments, on a small padded batch whose chroma tiles straddle three kernel bands, and with bilinear taps.
@pytest.mark.parametrize("ov,n,dims,pad", [
    (dict(interpolation_alg=CUBIC, num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1), 25, (3840, 1920, 1536, 1024), 0),
    (dict(interpolation_alg=CUBIC), 24, (3840, 1920, 1536, 1024), 0),                      # vf_transform360.c defaults: 5 x 1 segments
    (dict(interpolation_alg=CUBIC, num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1), 26, (960, 480, 384, 256), 64),
    (dict(interpolation_alg=LINEAR, num_vertical_segments=9, num_horizontal_segments=4), 24, (1920, 960, 768, 512), 0),
    (dict(interpolation_alg=CUBIC, fixed_yaw=33.0, fixed_pitch=-21.0, num_vertical_segments=15, num_horizontal_segments=8), 24, (1280, 640, 768, 512), 0),
])
def test_fused_lowpass_batches_equal_the_oracle(ov, n, dims, pad, T, oracle_mod):
    _batch_case(T, oracle_mod, dict(enable_multi_threading=1, **ov), n=n, dims=dims, extra_pad=pad, threads=_threads(), fused=True)


def test_fused_lowpass_pipelined_calls(T, oracle_mod):
    """the fused path through three pipeline lanes (per-lane scratch planes for the segments the unfused tiles still read)"""
    _batch_case(T, oracle_mod, dict(interpolation_alg=CUBIC, num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1),
                n=24, dims=(1920, 960, 768, 512), extra_pad=0, threads=_threads(), pipelined=True, fused=True)


def test_config1_batch_full_size_all_planes(T, oracle_mod):
    _batch_case(T, oracle_mod, dict(interpolation_alg=NEAREST, enable_low_pass_filter=0), n=33,
                dims=(1920, 960, 768, 512), extra_pad=0, threads=_threads())


def test_config4_batch_full_size_all_planes(T, oracle_mod):
    from tests.cases import TB
    _batch_case(T, oracle_mod, dict(interpolation_alg=LANCZOS4, enable_low_pass_filter=0, **TB), n=2,
                dims=(7680, 3840, 3072, 4096), extra_pad=0, threads=_threads())


def test_bilinear_batch_full_size_all_planes(T, oracle_mod):
    _batch_case(T, oracle_mod, dict(interpolation_alg=LINEAR, enable_low_pass_filter=0), n=26,
                dims=(3840, 1920, 1536, 1024), extra_pad=0, threads=_threads())


def test_batch_mixing_scaled_and_unscaled_planes(T, oracle_mod):
    """A scale factor can round one plane's map to a different size and leave the other's alone (1000 -> 1001 but
    500 -> 500): the batch call must take both branches."""
    import torch
    from transform360_amd import _lib
    ctx = filter_defaults(enable_low_pass_filter=0, width_scale_factor=1.0006, height_scale_factor=1.0)
    o = oracle_mod.Oracle(ctx, threads=4)
    dims = [(2048, 1024, 1000, 512), (1024, 512, 500, 256)]
    n_frames = 2
    in_bytes = sum(d[0] * d[1] for d in dims)
    out_bytes = sum(d[2] * d[3] for d in dims)
    d_in = torch.empty(n_frames * in_bytes, dtype=torch.uint8, device="cuda")
    T.fill_noise(d_in, 0xABCD)
    d_out = torch.zeros(n_frames * out_bytes, dtype=torch.uint8, device="cuda")
    descs = (_lib.T360PlaneDesc * 2)()
    io = oo = 0
    for k, (iw, ih, ow, oh) in enumerate(dims):
        descs[k] = _lib.T360PlaneDesc(in_offset=io, out_offset=oo, in_stride=iw, out_stride=ow, in_width=iw, in_height=ih,
                                      out_width=ow, out_height=oh, map_index=k)
        io += iw * ih
        oo += ow * oh
    with T.VideoFrameTransform(ctx) as t:
        for k, d in enumerate(dims):
            assert t.generateMapForPlane(*d, k) and o.generateMapForPlane(*d, k)
        assert t.map(0).shape[1] == 1001 and t.map(1).shape[1] == 500  # one plane resizes, the other does not
        assert t.setStream(torch.cuda.current_stream())
        assert t.transformFrames(d_in, in_bytes, d_out, out_bytes, n_frames, descs)
        assert t.synchronize()
    h_in, h_out = d_in.cpu().numpy(), d_out.cpu().numpy()
    for f in range(n_frames):
        io = oo = 0
        for k, (iw, ih, ow, oh) in enumerate(dims):
            src = h_in[f * in_bytes + io:][:iw * ih].reshape(ih, iw)
            want = np.zeros((oh, ow), np.uint8)
            assert o.transformFramePlane(src, want, k, k)
            got = h_out[f * out_bytes + oo:][:ow * oh].reshape(oh, ow)
            assert np.array_equal(got, want), (f, k)
            io += iw * ih
            oo += ow * oh


def test_five_planes_in_one_call(T, oracle_mod):
    """More planes than one fused launch holds (4): the batch is flushed in groups."""
    import torch
    in_w, in_h, out_w, out_h = 960, 480, 384, 256
    for interp in (CUBIC, LINEAR):
        ctx = filter_defaults(interpolation_alg=interp, enable_low_pass_filter=0)
        o = oracle_mod.Oracle(ctx, threads=4)
        n_planes, n_frames = 5, 3
        in_plane, out_plane = in_w * in_h, out_w * out_h
        d_in = torch.empty(n_frames * n_planes * in_plane, dtype=torch.uint8, device="cuda")
        T.fill_noise(d_in, 0x5EED)
        d_out = torch.zeros(n_frames * n_planes * out_plane, dtype=torch.uint8, device="cuda")
        from transform360_amd import _lib
        descs = (_lib.T360PlaneDesc * n_planes)()
        for k in range(n_planes):
            descs[k] = _lib.T360PlaneDesc(in_offset=k * in_plane, out_offset=k * out_plane, in_stride=in_w, out_stride=out_w,
                                          in_width=in_w, in_height=in_h, out_width=out_w, out_height=out_h, map_index=0)
        with T.VideoFrameTransform(ctx) as t:
            assert t.generateMapForPlane(in_w, in_h, out_w, out_h, 0) and o.generateMapForPlane(in_w, in_h, out_w, out_h, 0)
            assert t.setStream(torch.cuda.current_stream())
            assert t.transformFrames(d_in, n_planes * in_plane, d_out, n_planes * out_plane, n_frames, descs)
            assert t.synchronize()
        h_in, h_out = d_in.cpu().numpy(), d_out.cpu().numpy()
        for f in range(n_frames):
            for k in range(n_planes):
                src = h_in[(f * n_planes + k) * in_plane:][:in_plane].reshape(in_h, in_w)
                want = np.zeros((out_h, out_w), np.uint8)
                assert o.transformFramePlane(src, want, 0, k)
                got = h_out[(f * n_planes + k) * out_plane:][:out_plane].reshape(out_h, out_w)
                assert np.array_equal(got, want), (interp, f, k)


# ---------------------------------------------------------------- size-independent properties
@pytest.mark.parametrize("interp", [NEAREST, LINEAR, CUBIC, LANCZOS4])
def test_flat_plane_stays_flat(interp, T):
    """Every Q15 weight entry sums to exactly 32768 -> constant in, same constant out."""
    import torch
    with T.VideoFrameTransform(filter_defaults(interpolation_alg=interp, enable_low_pass_filter=0)) as t:
        assert t.generateMapForPlane(3840, 1920, 1536, 1024, 0)
        src = torch.full((1920, 3840), 173, dtype=torch.uint8, device="cuda")
        dst = torch.zeros((1024, 1536), dtype=torch.uint8, device="cuda")
        _ready()
        assert t.transformFramePlane(src, dst, 0)
        assert (dst == 173).all().item()


def test_nearest_full_size_is_a_gather_of_the_map(T):
    """Nearest output == input[round(map)] computed independently with torch indexing."""
    import torch
    with T.VideoFrameTransform(filter_defaults(interpolation_alg=NEAREST, enable_low_pass_filter=0)) as t:
        assert t.generateMapForPlane(3840, 1920, 1536, 1024, 0)
        m = torch.from_numpy(t.map(0)).cuda()
        src = torch.empty(1920 * 3840, dtype=torch.uint8, device="cuda")
        T.fill_noise(src, 77)
        src = src.view(1920, 3840)
        dst = torch.zeros((1024, 1536), dtype=torch.uint8, device="cuda")
        _ready()
        assert t.transformFramePlane(src, dst, 0)
        ix = torch.round(m[..., 0]).long() % 3840    # torch.round is round-half-even like cvRound
        iy = torch.round(m[..., 1]).long() % 1920
        assert torch.equal(dst, src[iy, ix])


# ---------------------------------------------------------------- error behaviour of the ABI
def test_error_convention(T, gpu_lib):
    L = gpu_lib
    L.VideoFrameTransform_delete(None)                       # delete(NULL) is a no-op (vf.c:334)
    assert L.VideoFrameTransform_new(None) is None
    with T.VideoFrameTransform(filter_defaults(enable_low_pass_filter=0)) as t:
        src = dev(np.zeros((64, 128), np.uint8))
        dst = dev(np.zeros((32, 48), np.uint8))
        _ready()
        assert not t.transformFramePlane(src, dst, 0)          # no map yet -> 0, no crash
        assert not t.generateMapForPlane(0, 64, 48, 32, 0)
        assert t.generateMapForPlane(128, 64, 48, 32, 0)
        _ready()
        assert t.transformFramePlane(src, dst, 0)
        _ready()
        assert not t.transformFramePlane(src, dst, 1)          # index 1 never generated
    # interpolation_alg = 3: the reference prints, writes nothing and still returns true (:780-783)
    with T.VideoFrameTransform(filter_defaults(interpolation_alg=3, enable_low_pass_filter=0)) as t:
        assert t.generateMapForPlane(128, 64, 48, 32, 0)
        dst = dev(np.full((32, 48), 9, np.uint8))
        _ready()
        assert t.transformFramePlane(dev(np.zeros((64, 128), np.uint8)), dst, 0)
        assert (dst == 9).all().item()


# ---------------------------------------------------------------- plain-C caller (ffmpeg stand-in)
def test_c_harness_replays_filter_sequence(T, oracle_mod, tmp_path):
    """tests/c/vf_sequence.c links against libTransform360.so like ffmpeg would
    (--extra-libs='-lTransform360 -lstdc++', reference README.md:67) and replays
    vf_transform360.c's call sequence with malloc'd frames and padded linesizes."""
    import os
    import subprocess

    from tests.conftest import ROOT
    from transform360_amd import _lib
    O = oracle_mod
    exe = tmp_path / "vf_sequence"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "vf_sequence.c"), "-o", str(exe), "-L", libdir,
                           "-lTransform360", "-lstdc++", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    in_w, in_h, edge, nframes = 1280, 640, 260, 2          # 260 -> 256 (multiple of 16)
    for interp, lowpass in ((CUBIC, 1), (NEAREST, 0)):
        out = subprocess.check_output([str(exe), str(in_w), str(in_h), str(edge), str(interp), str(lowpass),
                                       str(nframes)]).decode()
        lines = [l.split() for l in out.splitlines() if l.startswith("frame")]
        assert len(lines) == nframes * 3, out
        ctx = filter_defaults(interpolation_alg=interp, enable_low_pass_filter=lowpass, num_vertical_segments=15,
                              num_horizontal_segments=32)
        o = O.Oracle(ctx, threads=4)
        out_w, out_h = 768, 512
        cw, ch = chroma_dims(in_w, in_h)
        ocw, och = chroma_dims(out_w, out_h)
        assert o.generateMapForPlane(in_w, in_h, out_w, out_h, 0) and o.generateMapForPlane(cw, ch, ocw, och, 1)
        for l in lines:
            f, plane = int(l[1]), int(l[3])
            iw, ih, ow, oh = (in_w, in_h, out_w, out_h) if plane == 0 else (cw, ch, ocw, och)
            ls = (iw + 63) // 64 * 64 + 64
            seed = 0x360 ^ (f << 40) ^ (plane << 36)
            src = T.noise_bytes(ls * ih, seed).reshape(ih, ls)[:, :iw]
            want = np.zeros((oh, ow), np.uint8)
            assert o.transformFramePlane(src, want, 1 if plane else 0, plane)
            assert l[4] == "%dx%d" % (ow, oh)
            assert l[6] == hx(O.fnv1a64(want)), "frame %d plane %d differs from the oracle" % (f, plane)
            assert l[8] == "intact"


# ---------------------------------------------------------------- handle lifecycle / independence
def test_handles_are_independent_across_host_threads(T, oracle_mod):
    """one handle is used from one thread, DISTINCT handles must be independent.  Two host
    threads run their own handle (different interpolation, own stream) at the same time."""
    import threading

    import torch
    O = oracle_mod
    dims = (960, 480, 384, 256)
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, (dims[1], dims[0]), dtype=np.uint8)
    dsrc = dev(src)
    results, errors = {}, []

    def work(interp):
        try:
            ctx = filter_defaults(interpolation_alg=interp, enable_low_pass_filter=int(interp == CUBIC))
            with T.VideoFrameTransform(ctx) as t:
                assert t.generateMapForPlane(*dims, 0)
                outs = []
                for _ in range(20):
                    dst = torch.zeros((dims[3], dims[2]), dtype=torch.uint8, device="cuda")
                    torch.cuda.synchronize()
                    assert t.transformFramePlane(dsrc, dst, 0)
                    outs.append(dst.cpu().numpy())
                results[interp] = outs
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((interp, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in (CUBIC, LANCZOS4, NEAREST)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for interp, outs in results.items():
        o = O.Oracle(filter_defaults(interpolation_alg=interp, enable_low_pass_filter=int(interp == CUBIC)), threads=2)
        assert o.generateMapForPlane(*dims, 0)
        want = np.zeros((dims[3], dims[2]), np.uint8)
        assert o.transformFramePlane(src, want, 0)
        for got in outs:
            assert np.array_equal(got, want), interp


def test_handle_churn_does_not_leak_device_memory(T):
    import torch
    torch.cuda.synchronize()

    def cycle(n):
        for _ in range(n):
            with T.VideoFrameTransform(filter_defaults()) as t:
                assert t.generateMapForPlane(960, 480, 384, 256, 0)
                assert t.generateMapForPlane(480, 240, 192, 128, 1)
    cycle(3)
    free0, _ = torch.cuda.mem_get_info()
    cycle(25)
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "device memory shrank by %d MiB over 25 handle lifetimes" % ((free0 - free1) >> 20)


def test_native_multi_gpu_driver_matches_the_python_path(T):
    """examples/t360_multi_gpu.cpp (C++, links -lTransform360 -lrccl: one thread + handle + stream per device) on the
    devices of this box: its output checksum for device 0 equals the sum of the bytes the ctypes path produces for the
    same synthetic frames."""
    import subprocess
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "t360_multi_gpu")
    if not os.path.exists(exe):
        pytest.skip("examples/t360_multi_gpu not built (make -C examples; __graft_entry__.build() does it)")
    F = 5
    out = subprocess.run([exe, "--devices", "1", "--frames", str(F), "--steps", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = int(out.stdout.split("output checksum")[1].split()[0])
    ctx = filter_defaults(enable_low_pass_filter=0)
    lin, lout = T.FrameLayout(3840, 1920), T.FrameLayout(1536, 1024)
    with T.VideoFrameTransform(ctx) as t:
        for idx, k in ((0, 0), (1, 1)):
            assert t.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
        d_in = torch.empty(F * lin.frame_bytes, dtype=torch.uint8, device="cuda")
        for j in range(F):
            T.fill_noise(d_in[j * lin.frame_bytes:(j + 1) * lin.frame_bytes], T.frame_seed(j))
        d_out = torch.zeros(F * lout.frame_bytes, dtype=torch.uint8, device="cuda")
        _ready()
        assert t.transformFrames(d_in, lin.frame_bytes, d_out, lout.frame_bytes, F, t.plane_descs(lin, lout)) and t.synchronize()
        want = int(d_out.to(torch.int64).sum().item())
    assert got == want
    # the same frames through the pipelined calls with the gather path on: one worker has nobody to gather from, but RCCL is
    # initialised, the (empty) groups are posted and every step joins its lanes before the send would read the buffer
    out = subprocess.run([exe, "--workers", "1", "--frames", str(F), "--steps", "4", "--pipelined", "2", "--gather"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert int(out.stdout.split("output checksum")[1].split()[0]) == want
    # two workers sharing the device, BASELINE configs[4]-style sharding of 2 F frames: worker 0 owns frames [0, F)
    out = subprocess.run([exe, "--workers", "2", "--total-frames", str(2 * F), "--steps", "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert int(out.stdout.split("output checksum")[1].split()[0]) == want


@pytest.mark.parametrize("extra", [[], ["--pipelined", "2"]])
def test_native_driver_gathers_real_bytes_between_workers_on_one_device(extra):
    """examples/t360_multi_gpu --gather-local: three workers share this box's GPU and the outputs of
    workers 1 and 2 travel to worker 0's sink through the driver's own per-step operation lists, double buffering and
    events -- device copies standing in for ncclSend / ncclRecv, which RCCL refuses between ranks on one device -- for
    warm-up, timed and final steps; the sink must then hold exactly the bytes workers 1 and 2 computed."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "t360_multi_gpu")
    if not os.path.exists(exe):
        pytest.skip("examples/t360_multi_gpu not built (make -C examples; __graft_entry__.build() does it)")
    out = subprocess.run([exe, "--devices", "1", "--workers", "3", "--total-frames", "10", "--steps", "6", "--gather-local"] + extra,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gather check (device copies)" in out.stdout and out.stdout.rstrip().endswith(": ok"), out.stdout
    sums = [int(line.split("output checksum")[1].split()[0]) for line in out.stdout.splitlines() if "output checksum" in line]
    assert len(sums) == 3 and len(set(sums)) == 3      # three different shards of the stream
