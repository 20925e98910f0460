#!/bin/bash
# round 4, GPU call 5: persistent workgroups with padded ticket counters; lead of the ticket fetch 3 / 6 / 12 frames
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c5; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep64.txt
T360_LIB=$R/tools/ab/libT360_lead6.so tools/sweep.sh "T360_PERSIST=1" 2>&1 | tee -a $O/sweep64.txt
T360_LIB=$R/tools/ab/libT360_lead12.so tools/sweep.sh "T360_PERSIST=1" 2>&1 | tee -a $O/sweep64.txt
BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep8.txt
T360_LIB=$R/tools/ab/libT360_lead6.so BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_PERSIST=1" 2>&1 | tee -a $O/sweep8.txt
BENCH_EXTRA="--steps 10 --config 1" tools/sweep.sh "T360_PERSIST=1" "T360_PERSIST=0" 2>&1 | tee $O/sweep_cfg1.txt
BENCH_EXTRA="--steps 3 --warmup 1" tools/sweep.sh "T360_PERSIST=1 T360_TRACE=$O/trace64.bin" > /dev/null 2>&1
BENCH_EXTRA="--steps 3 --warmup 1 --frames 8" tools/sweep.sh "T360_PERSIST=1 T360_TRACE=$O/trace8.bin" > /dev/null 2>&1
python tools/trace_stats.py $O/trace64.bin | head -7
python tools/trace_stats.py $O/trace8.bin | head -7
