#!/bin/bash
# bicubic gather: staged s_waitcnt lgkmcnt per pixel (15 / 8 / 0) instead of one wait for the group's 32 reads
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1
for rep in 1 2 3; do for v in stw0 stw1; do
for args in "--config 2" "--config 2 --frames 8" "--config 3"; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py $args --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v | $args |', d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])"
done; done; done
