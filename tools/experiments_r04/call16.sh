#!/bin/bash
# round 4, GPU call 16: scatter tiles (4x4 output blocks grouped by their position in the SOURCE): parity, then time and
# fabric reads against the rectangular plan
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c16; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "instrumented" > $O/pytest.log 2>&1; grep -aE "passed|failed|Error" $O/pytest.log | tail -3
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_SCATTER=0" "T360_SCATTER=2" "T360_SCATTER=3" "T360_SCATTER=4" "T360_SCATTER=6" "T360_SCATTER=0" 2>&1 | tee $O/sweep.txt
for sc in 4; do
T360_SCATTER=$sc T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-host-abi --no-two-streams 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('scatter $sc verified', d['verified']['max_abs_diff'], d['verified']['differing_pixels'], d['gather_plan'][0])" | tee -a $O/sweep.txt
done
