import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transform360_amd import handler
from transform360_amd.abi import filter_defaults, CUBIC
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
ctx = filter_defaults(interpolation_alg=CUBIC, enable_low_pass_filter=0)
for rep in range(3):
    t0 = time.perf_counter(); t = handler.VideoFrameTransform(ctx); t1 = time.perf_counter()
    assert t.generateMapForPlane(3840, 1920, 1536, 1024, 0); t2 = time.perf_counter()
    assert t.generateMapForPlane(1920, 960, 768, 512, 1); t3 = time.perf_counter()
    print("handle %d: new %.1f ms, map 0 %.1f ms, map 1 %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    t.close()
