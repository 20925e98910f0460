#!/bin/bash
# output store cache policy of the tiled gather: 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
cd $R
for rep in 1 2 3; do
for v in base stnt st2 st3 st4 st5; do
for args in "--config 2 --frames 64"; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py $args --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v | $args |', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done; done
