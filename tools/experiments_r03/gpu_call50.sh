#!/bin/bash
# config 1 (768x512 output: 288 tiles of 128x16 for 512 workgroup slots): 8-wave plan vs 4-wave plan (576 tiles for 1 024 slots), frames per workgroup
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for rep in 1 2; do
for e in "T360_X=0" "T360_SMALL_BATCH=65" "T360_FRAMES_PER_BLOCK=32" "T360_SMALL_BATCH=65 T360_FRAMES_PER_BLOCK=32" "T360_FRAMES_PER_BLOCK=22"; do
 v=$(env $e python bench.py --config 1 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'][:27], 'verified', d['verified']['max_abs_diff'])")
 echo "$e: $v"
done; done
