"""Round 6: does the ADDRESS LAYOUT of the input batch (not its lines or bytes) move the hot kernel?  BASELINE config 2, 64 frames
per step, one stream, three batches rotating (2.1 GB), through T360_transformFrames with
  std          frames back to back (11 059 200 bytes apart: a multiple of 4 KiB)
  pad+N        the same with N more bytes between frames (other DRAM channel / bank phase per frame)
  interleaved  row r of frame f at (r * 64 + f) * stride: a tile's footprint of all 64 frames sits in ~12 MB of consecutive
               addresses (TLB reach), expressed through the plane strides alone
Prints ms per step (median of 5 x 20 steps).  (GPU box)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transform360_amd import handler as T, _lib
from transform360_amd.abi import CUBIC, filter_defaults

def run(name, frame_bytes, descs_of, total_bytes, F=64, steps=20, nbatch=3):
    ctx = filter_defaults(interpolation_alg=CUBIC, enable_low_pass_filter=0)
    lin, lout = T.FrameLayout(3840, 1920), T.FrameLayout(1536, 1024)
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    bufs = [torch.randint(0, 256, (total_bytes,), dtype=torch.uint8, device="cuda") for _ in range(nbatch)]
    d_out = torch.zeros(F * lout.frame_bytes, dtype=torch.uint8, device="cuda")
    with T.VideoFrameTransform(ctx) as t:
        for idx, k in ((0, 0), (1, 1)):
            assert t.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
        assert t.setStream(stream)
        descs = descs_of(t, lin, lout)
        def step(k):
            assert t.transformFrames(bufs[k % nbatch], frame_bytes, d_out, lout.frame_bytes, F, descs)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            for k in range(8): step(k)
            torch.cuda.synchronize()
        res = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(steps): step(k)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / steps * 1e3)
        print("%-14s %.4f ms per step  %s  kernel %s" % (name, sorted(res)[2], ["%.4f" % r for r in res], t.lastKernel()))
    del bufs

lin = T.FrameLayout(3840, 1920)
std = lambda t, li, lo: t.plane_descs(li, lo)
run("std", lin.frame_bytes, std, 64 * lin.frame_bytes)
for pad in (256, 2304, 4096 + 256, 65536 + 1280):
    run("pad+%d" % pad, lin.frame_bytes + pad, std, 64 * (lin.frame_bytes + pad))
def inter(t, li, lo):
    arr = (_lib.T360PlaneDesc * 3)()
    rowpitch = 64 * 3840
    offs = [0, 1920 * rowpitch, 1920 * rowpitch + 960 * rowpitch]
    for k in range(3):
        arr[k] = _lib.T360PlaneDesc(in_offset=offs[k], out_offset=lo.offsets[k], in_stride=rowpitch, out_stride=lo.strides[k],
                                   in_width=li.dims[k][0], in_height=li.dims[k][1], out_width=lo.dims[k][0], out_height=lo.dims[k][1],
                                   map_index=1 if k else 0)
    return arr
run("interleaved", 3840, inter, (1920 + 960 + 960) * 64 * 3840 + 64 * 3840)
