"""Extended soak of the fused low-pass path (GPU box): tests/test_gpu_fuzz.py::test_random_lowpass_batches_with_the_fused_path
over seeds beyond the suite's.  usage: python tests/soak/fused_soak.py [first] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle import t360_oracle as O  # noqa: E402
import tests.test_gpu_fuzz as F  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 300)
t0, ran = time.time(), 0
for seed in range(first, first + count):
    F._FUSED_RAN.clear()
    F.test_random_lowpass_batches_with_the_fused_path.__wrapped__(seed, O) if hasattr(F.test_random_lowpass_batches_with_the_fused_path, "__wrapped__") else F.test_random_lowpass_batches_with_the_fused_path(seed, O)
    ran += int(F._FUSED_RAN[-1])
print("fused soak: seeds %d..%d bit-exact, %d of %d ran remap_fused_kernel, %.0f s" % (first, first + count - 1, ran, count, time.time() - t0))
