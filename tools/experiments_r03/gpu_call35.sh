#!/bin/bash
# balanced tail runs (T360_X=1: the old 16 + remainder split), config 2, frames per step 17..48
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for f in 17 20 23 24 28 33 40 47 56 64; do
 a=$(T360_X=1 python bench.py --config 2 --frames $f --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['strong_cfg5']['ms_per_step'])")
 b=$(python bench.py --config 2 --frames $f --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['strong_cfg5']['ms_per_step'], 'verified', d['verified']['max_abs_diff'])")
 echo "frames $f: old split $a | balanced $b"
done
