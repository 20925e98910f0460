#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "lowpass or lpf or config3 or cfg3 or filter" 2>&1 | tail -2
for i in 1 2; do
T360_LIB=$R/tools/ab/libT360_old.so T360_BENCH_ALLOW_INSTRUMENTED=1 python bench.py --config 3 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old cfg3', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --config 3 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new cfg3', d['ms_per_step'], d['roofline']['frac'], d['verified']['max_abs_diff'])"
done
