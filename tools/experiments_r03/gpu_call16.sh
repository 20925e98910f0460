#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
for i in 1 2; do
T360_LIB=$R/tools/ab/libT360_old.so T360_BENCH_ALLOW_INSTRUMENTED=1 python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['host_abi'])"
python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['host_abi'])"
done
