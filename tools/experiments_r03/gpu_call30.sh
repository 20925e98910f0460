#!/bin/bash
# s_setprio around the phases of the frame loop: 1 store + DMA issue high, 2 everything but the gather high, 3 the gather high
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
cd $R
for rep in 1 2 3; do
for v in base prio1 prio2 prio3; do
for args in "--config 2 --frames 64" "--config 2 --frames 8"; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py $args --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v | $args |', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done; done
