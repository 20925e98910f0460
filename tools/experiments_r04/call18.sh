#!/bin/bash
# round 4, GPU call 18: scatter tiles with write-back (not streaming) output stores: do partial lines merge in the L2?
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c18; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 12"
T360_LIB=$R/tools/ab/libT360_wb.so tools/sweep.sh "T360_SCATTER=0" "T360_SCATTER=1" "T360_SCATTER=2" "T360_SCATTER=4" 2>&1 | tee $O/sweep.txt
tools/sweep.sh "T360_SCATTER=1" "T360_SCATTER=0" 2>&1 | tee -a $O/sweep.txt
