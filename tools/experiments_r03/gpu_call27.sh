#!/bin/bash
# cache policy of the staging loads / output stores of the tiled gather: A/B of four builds (tools/ab_build.sh), config 2 + 8 frames + config 1/4
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
cd $R
for rep in 1 2; do
for v in base ldnt stnt bothnt; do
for args in "--config 2 --frames 64" "--config 2 --frames 8" "--config 1 --frames 64" "--config 4 --frames 64 --steps 5 --warmup 2"; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py $args --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v | $args |', d['ms_per_step'], d['roofline']['avg_launch_ms'], 'verified', d['verified']['max_abs_diff'])"
done; done; done
