#!/bin/bash
# tile tables read with non-temporal loads (they are read once per launch and should not push source lines out of the L2)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1
for rep in 1 2 3; do for v in base tabnt; do
for args in "--config 2 --frames 64" "--config 2 --frames 8"; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py $args --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v | $args |', d['strong_cfg5']['ms_per_step'], d['roofline']['avg_launch_ms'], 'verified', d['verified']['max_abs_diff'])"
done; done; done
