#!/bin/bash
# launches with fewer tiles than workgroup slots split the frames into runs: config 1 and the others, before (tools/ab/libT360_base.so) / after
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/suite_pytest.log 2>&1; grep -aE "passed|failed" gpurun_out/suite_pytest.log
export T360_BENCH_ALLOW_INSTRUMENTED=1
for args in "--config 1" "--config 1 --frames 8" "--config 1 --frames 16" "--config 2" "--config 2 --frames 8"; do
 a=$(T360_LIB=$R/tools/ab/libT360_base.so python bench.py $args --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])")
 b=$(T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so python bench.py $args --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])")
 echo "$args: before $a | after $b"
done
