#!/bin/bash
# Lanczos4 plan: bank-aware row placement with per-tile skew search (default) vs rows packed back to back (T360_ROW_ALIGN=1):
# kernel time and first-step (planning) latency, config 4
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for e in "T360_X=0" "T360_ROW_ALIGN=1" "T360_X=0" "T360_ROW_ALIGN=1"; do
env $e python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'], 'first step', d['first_step_ms'], 'verified', d['verified']['max_abs_diff'])"
done
