#!/bin/bash
# config 3: pitch of the filtered chroma scratch planes 2048 (T360_X=1) vs 2304 bytes
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for rep in 1 2 3; do
 a=$(T360_X=1 python bench.py --config 3 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])")
 b=$(python bench.py --config 3 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])")
 echo "pitch 2048: $a | pitch 2304: $b"
done
