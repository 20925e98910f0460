#!/bin/bash
# Lanczos4 gather: all 24 stencil dwords read up front, rows consumed as they arrive (T360_LZ_ALL) vs two halves
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1
for rep in 1 2 3; do for v in base lzall; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v |', d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])"
done; done
