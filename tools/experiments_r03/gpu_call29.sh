#!/bin/bash
# low-pass kernel: non-temporal source loads / output stores (config 3)
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
cd $R
for rep in 1 2 3; do
for v in base lpld lpst lpboth; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py --config 3 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v |', d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])"
done; done
