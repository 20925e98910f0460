#!/usr/bin/env python3
"""tests/soak/fuzz_soak.py [first_seed] [count] [plane|batch|pipe|plane4|tiny] -- the random configurations of tests/test_gpu_fuzz.py (single
planes, or yuv420p batches) for seeds beyond the ones the suite pins (development soak; prints every mismatch, exits 1
if there was one)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import t360_oracle as O  # noqa: E402
from tests import test_gpu_fuzz as F  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 150
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mode = sys.argv[3] if len(sys.argv) > 3 else "plane"
fn = F.test_random_batches_match_oracle if mode == "batch" else F.test_random_configuration_matches_oracle
if mode == "pipe":   # random batches, plain and through three pipelined lanes
    fn = lambda seed, oracle: F.test_random_batches_match_oracle(seed, oracle, pipelined=True)  # noqa: E731
if mode == "tiny":
    # the same configurations on very small planes (1 .. 40 px a side, any alignment)
    small = F.draw

    def tiny(seed):
        import numpy as np
        ov, _, pin, pout = small(seed)
        r = np.random.default_rng(seed ^ 0x5EED)
        return ov, (int(r.integers(1, 41)), int(r.integers(1, 41)), int(r.integers(1, 41)), int(r.integers(1, 41))), pin, pout
    F.draw = tiny
if mode == "plane4":
    # the same configurations on planes four times as wide and high (all tile shapes of the gather plans appear)
    small = F.draw

    def big(seed):
        ov, (iw, ih, ow, oh), pin, pout = small(seed)
        return ov, (iw * 4, ih * 4, ow * 4, oh * 4), pin, pout
    F.draw = big
O.build(ref=False)
bad = 0
for seed in range(first, first + count):
    try:
        fn(seed, O)
    except AssertionError as e:
        bad += 1
        print("seed %d: %s" % (seed, str(e)[:1500]), flush=True)
print("seeds %d..%d: %d mismatches" % (first, first + count - 1, bad))
sys.exit(1 if bad else 0)
