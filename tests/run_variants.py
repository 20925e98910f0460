#!/usr/bin/env python3
"""Child process of tests/test_gpu_parity.py::test_instrumented_build_variants_are_bit_identical: with T360_LIB set to
the instrumented library, run a small yuv420p batch through T360_transformFrames once per environment setting and
compare every plane with the oracle (the switches are read when a handle is created)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    variants = json.loads(sys.argv[1])
    from oracle import t360_oracle as O
    from tests.test_gpu_parity import _batch_case
    from transform360_amd import _lib, handler
    assert _lib.load().T360_buildFlags() != 0, "T360_LIB must point at the instrumented build"
    keys = sorted({k for v in variants for k in v})
    for v in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ["T360_SMALL_BATCH"] = "0"   # the 5-frame batch below must exercise the plan the variant configures
        os.environ.update(v)
        ov = dict(num_vertical_segments=5, num_horizontal_segments=4) if ("T360_NO_FAST_LOWPASS" in v or "T360_NO_WIDE_LOWPASS" in v) else dict(
            enable_low_pass_filter=0)
        _batch_case(handler, O, ov, n=5, extra_pad=0)
        print("ok", v, flush=True)
    print("variants ok: %d" % len(variants))


if __name__ == "__main__":
    main()
