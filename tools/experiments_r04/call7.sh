#!/bin/bash
# round 4, GPU call 7: cache policy of the staging loads (L1 bypass), tail knobs with the cheaper prologue, occupancy
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c7; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_X=base" 2>&1 | tee $O/sweep.txt
for v in sc1 sc0sc1; do T360_LIB=$R/tools/ab/libT360_$v.so tools/sweep.sh "T360_X=$v" 2>&1 | tee -a $O/sweep.txt; done
tools/sweep.sh "T360_TAIL_PCT=20" "T360_TAIL_PCT=30" "T360_TAIL_PCT=20 T360_TAIL_FRAMES=8" "T360_TAIL_PCT=6" "T360_LDS_PAD=8192" "T360_SMALL_BATCH=1000" "T360_MAX_PIECES=16" "T360_X=base_again" 2>&1 | tee -a $O/sweep.txt
BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_X=base" "T360_TAIL_PCT=30" "T360_TAIL_PCT=50" "T360_TAIL_PCT=0" 2>&1 | tee $O/sweep8.txt
for v in sc1; do T360_LIB=$R/tools/ab/libT360_$v.so BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_X=$v" 2>&1 | tee -a $O/sweep8.txt; done
