#!/bin/bash
# round 4, GPU call 3: the shrunk weight table (2 loads per pixel) + DMA-first prologue: 64-frame and 8-frame steps, and the
# per-workgroup timeline of both (T360_TRACE, tools/trace_stats.py)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c3; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_X=new" 2>&1 | tee $O/sweep64.txt
BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_X=new" 2>&1 | tee $O/sweep8.txt
T360_LIB=$R/tools/ab/libT360_old.so tools/sweep.sh "T360_X=old" 2>&1 | tee -a $O/sweep64.txt
T360_LIB=$R/tools/ab/libT360_old.so BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_X=old" 2>&1 | tee -a $O/sweep8.txt
BENCH_EXTRA="--steps 3 --warmup 1" tools/sweep.sh "T360_TRACE=$O/trace64.bin" > /dev/null 2>&1
BENCH_EXTRA="--steps 3 --warmup 1 --frames 8" tools/sweep.sh "T360_TRACE=$O/trace8.bin" > /dev/null 2>&1
python tools/trace_stats.py $O/trace64.bin | tee $O/trace64.txt
python tools/trace_stats.py $O/trace8.bin | tee $O/trace8.txt
tools/sweep.sh "T360_X=new_again" 2>&1 | tee -a $O/sweep64.txt
