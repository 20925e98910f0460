#!/bin/bash
# round 4, GPU call 22: blocked filtered planes, 64x2 blocks (default build) vs 32x4 vs row-major; per-kernel times
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c22; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 10 --config 3"
tools/sweep.sh "T360_BLOCKED=1" "T360_BLOCKED=0" 2>&1 | tee $O/sweep.txt
T360_LIB=$R/tools/ab/libT360_blk32.so tools/sweep.sh "T360_BLOCKED=1" 2>&1 | tee -a $O/sweep.txt
tools/sweep.sh "T360_BLOCKED=1" "T360_BLOCKED=0" 2>&1 | tee -a $O/sweep.txt
cd /tmp && export TMPDIR=/tmp
for b in 1 0; do
  T360_BLOCKED=$b T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$b -o t -- \
    python $R/bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-host-abi > $O/t$b.log 2>&1
  echo "== blocked $b: $(grep -o '"max_abs_diff": [0-9]*' $O/t$b.log | head -1)"; head -3 $O/t$b/t_kernel_stats.csv | cut -d, -f2-4
done 2>&1 | tee -a $O/sweep.txt
