#!/bin/bash
# config 3: long low-pass batches as two half batches on two streams (T360_LP_SPLIT 0 off, 1 together, 2 staggered)
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
cd $R
for f in 64 48; do
for e in "T360_LP_SPLIT=0" "T360_LP_SPLIT=1" "T360_LP_SPLIT=2" "T360_LP_SPLIT=0" "T360_LP_SPLIT=1" "T360_LP_SPLIT=2"; do
env $e python bench.py --config 3 --frames $f --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $f $e', d['ms_per_step'], d['repeats_ms_per_step'], 'verified', d['verified']['max_abs_diff'])"
done; done
