#!/bin/bash
# round 4, GPU call 15: the packed weight table staged into LDS once per workgroup (8-wave bilinear / bicubic) vs gathered from
# global memory per pixel; parity first
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c15; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -aE "passed|failed" $O/pytest.log | tail -2
export BENCH_EXTRA="--steps 20"
for rep in 1 2; do
tools/sweep.sh "T360_X=ldstable" 2>&1 | tee -a $O/sweep.txt
T360_LIB=$R/tools/ab/libT360_notable.so tools/sweep.sh "T360_X=global_gather" 2>&1 | tee -a $O/sweep.txt
done
for fr in 16 32; do
BENCH_EXTRA="--steps 20 --frames $fr" tools/sweep.sh "T360_SMALL_BATCH=0" 2>&1 | tee -a $O/sweep_short.txt
T360_LIB=$R/tools/ab/libT360_notable.so BENCH_EXTRA="--steps 20 --frames $fr" tools/sweep.sh "T360_SMALL_BATCH=0" "T360_X=default_plan" 2>&1 | tee -a $O/sweep_short.txt
done
BENCH_EXTRA="--steps 20 --frames 8" tools/sweep.sh "T360_SMALL_BATCH=0" "T360_X=default_plan" 2>&1 | tee -a $O/sweep_short.txt
BENCH_EXTRA="--steps 10 --config 3" tools/sweep.sh "T360_X=ldstable" 2>&1 | tee -a $O/sweep.txt
T360_LIB=$R/tools/ab/libT360_notable.so BENCH_EXTRA="--steps 10 --config 3" tools/sweep.sh "T360_X=global_gather" 2>&1 | tee -a $O/sweep.txt
