#!/usr/bin/env python3
"""Determinism soak: the same batch transformed N times must give byte-identical output every time
(a missed DMA wait or a ring race would show up as a rare difference).  usage: soak.py [config] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from transform360_amd import handler  # noqa: E402
from transform360_amd.abi import config_output, filter_defaults  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wl = bench.workload(cfg)
ctx = filter_defaults(**wl["ov"])
in_w, in_h = wl["in_w"], wl["in_h"]
out_w, out_h = config_output(in_w, in_h, wl["edge"], ctx.output_layout, ctx.input_stereo_format, ctx.output_stereo_format)
lin, lout = handler.FrameLayout(in_w, in_h), handler.FrameLayout(out_w, out_h)
F = 16 if cfg == 4 else 64
t = handler.VideoFrameTransform(ctx)
for idx, k in ((0, 0), (1, 1)):
    assert t.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
assert t.setStream(torch.cuda.current_stream())
d_in = torch.empty(F * lin.frame_bytes, dtype=torch.uint8, device="cuda")
for j in range(F):
    handler.fill_noise(d_in[j * lin.frame_bytes:(j + 1) * lin.frame_bytes], handler.frame_seed(j))
descs = t.plane_descs(lin, lout)
ref = torch.zeros(F * lout.frame_bytes, dtype=torch.uint8, device="cuda")
assert t.transformFrames(d_in, lin.frame_bytes, ref, lout.frame_bytes, F, descs)
torch.cuda.synchronize()
bad = 0
for i in range(iters):
    out = torch.zeros_like(ref)
    assert t.transformFrames(d_in, lin.frame_bytes, out, lout.frame_bytes, F, descs)
    if not torch.equal(out, ref):
        bad += 1
        print("iteration %d differs in %d bytes" % (i, int((out != ref).sum().item())))
# the same batch through the pipelined calls (T360_transformFramesPipelined, three lanes, three output buffers): overlapping
# launches on different streams must not disturb each other either
assert t.setPipelineDepth(3)
outs = [torch.zeros_like(ref) for _ in range(3)]
pbad = 0
piters = max(3, iters // 3)
for i in range(piters):
    assert t.transformFramesPipelined(d_in, lin.frame_bytes, outs[i % 3], lout.frame_bytes, F, descs)
    if i % 3 == 2:
        assert t.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                pbad += 1
            o.zero_()
        torch.cuda.synchronize()
assert t.synchronize()
bad += pbad
print("config %d: %d iterations x %d frames, %d differing (of which pipelined calls: %d iterations, %d differing)" % (cfg, iters, F, bad, piters, pbad))
t.close()
sys.exit(1 if bad else 0)
