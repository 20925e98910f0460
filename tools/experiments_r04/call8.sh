#!/bin/bash
# round 4, GPU call 8: config 1 (nearest, small frames): 4-wave workgroups (4 per CU) against the 8-wave default, run lengths
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c8; mkdir -p $O
cd $R
BENCH_EXTRA="--steps 20 --config 1" tools/sweep.sh "T360_X=base" "T360_SMALL_BATCH=1000" "T360_SMALL_BATCH=1000 T360_TAIL_PCT=0" "T360_FRAMES_PER_BLOCK=16" "T360_FRAMES_PER_BLOCK=64 T360_TAIL_PCT=0" "T360_SMALL_BATCH=1000 T360_FRAMES_PER_BLOCK=32" "T360_X=base" 2>&1 | tee $O/sweep_cfg1.txt
BENCH_EXTRA="--steps 3 --warmup 1 --config 1" tools/sweep.sh "T360_TRACE=$O/trace_cfg1.bin" > /dev/null 2>&1
python tools/trace_stats.py $O/trace_cfg1.bin | head -8
