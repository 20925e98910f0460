#!/bin/bash
# round 4, GPU call 13: L2 pre-touch of the frame D frames ahead (one dword load per 128-byte line, result discarded)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c13; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_X=base" 2>&1 | tee $O/sweep.txt
for v in pt2 pt3 pt4; do T360_LIB=$R/tools/ab/libT360_$v.so tools/sweep.sh "T360_X=$v" "T360_DEBUG=1" 2>&1 | tee -a $O/sweep.txt; done
tools/sweep.sh "T360_X=base" "T360_DEBUG=1" 2>&1 | tee -a $O/sweep.txt
T360_LIB=$R/tools/ab/libT360_pt3.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-host-abi --no-two-streams 2>&1 | tail -1 | grep -o '"verified": {[^}]*}' | tee -a $O/sweep.txt
