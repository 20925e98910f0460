"""development: where does the fused low-pass + gather differ from the oracle?  (GPU box)"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import t360_oracle as O
from transform360_amd import handler as T
from transform360_amd.abi import CUBIC, filter_defaults

def run(in_w, in_h, out_w, out_h, n, **ov):
    ctx = filter_defaults(interpolation_alg=CUBIC, **ov)
    lin, lout = T.FrameLayout(in_w, in_h), T.FrameLayout(out_w, out_h)
    d_in = torch.empty(n * lin.frame_bytes, dtype=torch.uint8, device="cuda")
    for k in range(n):
        T.fill_noise(d_in[k * lin.frame_bytes:(k + 1) * lin.frame_bytes], T.frame_seed(k))
    d_out = torch.full((n * lout.frame_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    o = O.Oracle(ctx, threads=16)
    ctx2 = filter_defaults(interpolation_alg=CUBIC, **dict(ov, enable_low_pass_filter=0))
    o2 = O.Oracle(ctx2, threads=16)
    with T.VideoFrameTransform(ctx) as t:
        for idx, k in ((0, 0), (1, 1)):
            d = (*lin.dims[k], *lout.dims[k])
            assert t.generateMapForPlane(*d, idx) and o.generateMapForPlane(*d, idx) and o2.generateMapForPlane(*d, idx)
        assert t.setStream(torch.cuda.current_stream())
        assert t.transformFrames(d_in, lin.frame_bytes, d_out, lout.frame_bytes, n, t.plane_descs(lin, lout))
        assert t.synchronize()
        print("kernel:", t.lastKernel())
    for k in (0, 1, n - 1):
        fin = d_in[k * lin.frame_bytes:(k + 1) * lin.frame_bytes].cpu().numpy()
        fout = d_out[k * lout.frame_bytes:(k + 1) * lout.frame_bytes].cpu().numpy()
        for p in range(3):
            want = np.zeros((lout.dims[p][1], lout.dims[p][0]), np.uint8)
            raw = np.zeros_like(want)
            assert o.transformFramePlane(lin.plane_view(fin, p), want, 1 if p else 0, p)
            assert o2.transformFramePlane(lin.plane_view(fin, p), raw, 1 if p else 0, p)
            got = lout.plane_view(fout, p)
            bad = got != want
            print("frame %d plane %d: %d of %d differ; equal to the UNFILTERED gather on %d of the differing; untouched (0x5A) %d; max|d| %d" % (
                k, p, bad.sum(), bad.size, (bad & (got == raw)).sum(), (bad & (got == 0x5A)).sum(), np.abs(got.astype(int) - want.astype(int)).max()))
            if bad.any() and p <= 1 and k == 0:
                # 16-row x 128-col tile map of mismatches
                h, w = bad.shape
                tiles = bad.reshape(h // 16, 16, w // 16, 16).sum(axis=(1, 3))
                for row in tiles[: 64]:
                    print(" ".join("%3d" % v if v else "  ." for v in row))
                ys, xs = np.nonzero(bad)
                print("first mismatches:", [(int(x), int(y), int(got[y, x]), int(want[y, x]), int(raw[y, x])) for x, y in list(zip(xs, ys))[:12]])

if __name__ == "__main__":
    cfg3 = dict(enable_low_pass_filter=1, num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1)
    if len(sys.argv) > 1:
        run(*[int(v) for v in sys.argv[1:6]], **cfg3)
    else:
        run(960, 480, 384, 256, 25, **cfg3)
        run(480, 240, 192, 128, 25, **cfg3)
