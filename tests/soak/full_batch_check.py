#!/usr/bin/env python3
"""tests/soak/full_batch_check.py -- ALL 64 frames of the benchmarked batch (BASELINE configs 2 and 3, full size, yuv420p,
one T360_transformFrames call) against per-plane oracle calls.  The suite checks 17-33 frames of each configuration and
bench.py six frames per run; this is the one-off complete comparison."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import t360_oracle as O  # noqa: E402
from tests.test_gpu_parity import _batch_case  # noqa: E402
from transform360_amd import handler as T  # noqa: E402
from transform360_amd.abi import CUBIC  # noqa: E402

for name, ov in (("config 2", dict(interpolation_alg=CUBIC, enable_low_pass_filter=0)),
                 ("config 3", dict(interpolation_alg=CUBIC, num_vertical_segments=15, num_horizontal_segments=32, adjust_kernel=1))):
    t0 = time.time()
    _batch_case(T, O, ov, n=64, dims=(3840, 1920, 1536, 1024), extra_pad=0, threads=32)
    print("%s: 64 frames x 3 planes bit-identical to the oracle (%.0f s)" % (name, time.time() - t0), flush=True)
