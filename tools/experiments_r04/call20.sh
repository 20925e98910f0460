#!/bin/bash
# round 4, GPU call 20: filtered planes in 32x4-pixel blocks (config 3): parity, time, traffic
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c20; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -aE "passed|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export BENCH_EXTRA="--steps 10 --config 3"
tools/sweep.sh "T360_BLOCKED=1" "T360_BLOCKED=0" "T360_BLOCKED=1" "T360_BLOCKED=0" 2>&1 | tee $O/sweep.txt
T360_BLOCKED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python bench.py --config 3 --steps 5 --no-cpu-baseline --no-host-abi 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('blocked verified', d['verified']['max_abs_diff'], d['verified']['differing_pixels'], d['ms_per_step'])" | tee -a $O/sweep.txt
