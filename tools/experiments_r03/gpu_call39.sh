#!/bin/bash
# what would a low-pass fused into the gather's staging cost?  fake filter pass (T360_FAKE_FILTER) in the bicubic kernel, config 2
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1
for rep in 1 2; do for v in base fakef; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py --config 2 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v |', d['ms_per_step'], d['roofline']['avg_launch_ms'], 'verified', d['verified']['max_abs_diff'])"
done; done
