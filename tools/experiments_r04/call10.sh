#!/bin/bash
# round 4, GPU call 10: shorter runs per tile with the cheaper prologue (drift per run halves; tile-major order)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c10; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_X=base" "T360_FRAMES_PER_BLOCK=32" "T360_FRAMES_PER_BLOCK=32 T360_TAIL_PCT=0" "T360_FRAMES_PER_BLOCK=22" "T360_FRAMES_PER_BLOCK=16 T360_TAIL_PCT=0" "T360_X=base" 2>&1 | tee $O/sweep.txt
