#!/bin/bash
# round 4, GPU call 4: persistent workgroups with per-XCD ticket queues vs one item per workgroup (same build, T360_PERSIST)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c4; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep64.txt
BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep8.txt
BENCH_EXTRA="--steps 10 --config 1" tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep_cfg1.txt
BENCH_EXTRA="--steps 5 --config 4" tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep_cfg4.txt
BENCH_EXTRA="--steps 10 --config 3" tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep_cfg3.txt
BENCH_EXTRA="--steps 3 --warmup 1" tools/sweep.sh "T360_PERSIST=1 T360_TRACE=$O/trace64.bin" > /dev/null 2>&1
python tools/trace_stats.py $O/trace64.bin | tee $O/trace64.txt
