#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so BENCH_EXTRA="--config 4"
cd $R
for e in "T360_X=0" "T360_ROW_ALIGN=1" "T360_X=0" "T360_ROW_ALIGN=1"; do
env $e python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'], 'first step', d['first_step_ms'])"
done
