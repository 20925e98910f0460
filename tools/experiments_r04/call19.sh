#!/bin/bash
# round 4, GPU call 19: does the scatter-capable kernel (option off) cost the default path anything?  same box, alternating
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c19; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
for rep in 1 2 3; do
T360_LIB=$R/tools/ab/libT360_head.so tools/sweep.sh "T360_X=head" 2>&1 | tee -a $O/sweep.txt
tools/sweep.sh "T360_X=scatter_capable" 2>&1 | tee -a $O/sweep.txt
done
