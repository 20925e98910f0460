#!/bin/bash
# round 4, GPU call 11: dword stores for the one-pixel-per-lane tiles (Lanczos4, pole tiles): parity + config 4 / 2 A/B
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c11; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
BENCH_EXTRA="--steps 6 --config 4" tools/sweep.sh "T360_X=new" 2>&1 | tee $O/sweep.txt
T360_LIB=$R/tools/ab/libT360_prestore.so BENCH_EXTRA="--steps 6 --config 4" tools/sweep.sh "T360_X=old" 2>&1 | tee -a $O/sweep.txt
BENCH_EXTRA="--steps 6 --config 4" tools/sweep.sh "T360_X=new" 2>&1 | tee -a $O/sweep.txt
BENCH_EXTRA="--steps 20" tools/sweep.sh "T360_X=new" 2>&1 | tee -a $O/sweep.txt
T360_LIB=$R/tools/ab/libT360_prestore.so BENCH_EXTRA="--steps 20" tools/sweep.sh "T360_X=old" 2>&1 | tee -a $O/sweep.txt
