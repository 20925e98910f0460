#!/bin/bash
# round 4, GPU call 9: ring depth of the nearest-neighbour kernel (3 = before, 4, 6 = new default, 8), 8- and 4-wave workgroups
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c9; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20 --config 1"
for v in near3 near4 near8; do T360_LIB=$R/tools/ab/libT360_$v.so tools/sweep.sh "T360_X=$v" "T360_SMALL_BATCH=1000" 2>&1 | tee -a $O/sweep_cfg1.txt; done
tools/sweep.sh "T360_X=near6" "T360_SMALL_BATCH=1000" "T360_FRAMES_PER_BLOCK=64 T360_TAIL_PCT=0" 2>&1 | tee -a $O/sweep_cfg1.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nearest or config1" 2>&1 | tail -2
