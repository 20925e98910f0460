#!/bin/bash
# round 4, GPU call 6: same-box A/B of the committed kernel (one item per workgroup, no ticket loop in the code) against the
# persistent build run both ways
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c6; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
for rep in 1 2; do
T360_LIB=$R/tools/ab/libT360_prepersist.so tools/sweep.sh "T360_X=prepersist" 2>&1 | tee -a $O/sweep64.txt
tools/sweep.sh "T360_PERSIST=0" "T360_PERSIST=1" 2>&1 | tee -a $O/sweep64.txt
done
T360_LIB=$R/tools/ab/libT360_prepersist.so BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_X=prepersist" 2>&1 | tee -a $O/sweep8.txt
BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_PERSIST=0" "T360_PERSIST=1" 2>&1 | tee -a $O/sweep8.txt
