#!/bin/bash
# round-3 soak of the final build: determinism, random configurations beyond the suite's seeds, the complete 64-frame batches
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/soak; mkdir -p $O
cd $R
{
sha256sum transform360_amd/lib/libTransform360.so | cut -c1-16
for c in 2 3 1; do timeout 200 python tools/soak.py $c 300; done
timeout 200 python tools/soak.py 4 40
timeout 300 python tests/soak/full_batch_check.py
timeout 240 python tests/soak/fuzz_soak.py 7000 600 plane 2>&1 | tail -3
timeout 240 python tests/soak/fuzz_soak.py 7000 200 batch 2>&1 | tail -3
timeout 240 python tests/soak/fuzz_soak.py 7000 150 plane4 2>&1 | tail -3
timeout 120 python tests/soak/fuzz_soak.py 7000 600 tiny 2>&1 | tail -3
} 2>&1 | grep -v "^Could not\|amdgpu.ids" | tee $O/soak.txt
