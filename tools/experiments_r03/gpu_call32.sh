#!/bin/bash
# low-pass staging index math: suite + config 3 before/after (base = tools/ab/libT360_base.so built before the change)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "lowpass or low_pass or filter or cfg3 or config3 or batch or fuzz" > gpurun_out/lp_pytest.log 2>&1; grep -aE "passed|failed" gpurun_out/lp_pytest.log
export T360_BENCH_ALLOW_INSTRUMENTED=1
for rep in 1 2 3; do
for v in $R/tools/ab/libT360_base.so $R/transform360_amd/lib/libTransform360_instr.so; do
T360_LIB=$v python bench.py --config 3 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $v) |', d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])"
done; done
