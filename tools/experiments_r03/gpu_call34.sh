#!/bin/bash
# pipelined wide low-pass kernel: frames per workgroup sweep (config 3), vs the one-frame-per-workgroup kernel (tools/ab/libT360_base.so)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "lowpass or low_pass or filter or batch or fuzz" > gpurun_out/lp_pytest.log 2>&1; grep -aE "passed|failed" gpurun_out/lp_pytest.log
export T360_BENCH_ALLOW_INSTRUMENTED=1
for rep in 1 2; do
T360_LIB=$R/tools/ab/libT360_base.so python bench.py --config 3 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base |', d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])"
for fpb in 1 2 4 8 16 32; do
T360_LP_FPB=$fpb T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so python bench.py --config 3 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fpb $fpb |', d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])"
done; done
