#!/bin/bash
# round 4, GPU call 21: blocked filtered planes: which kernel pays?  kernel trace of config 3, both ways
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c21; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 1 0; do
  T360_BLOCKED=$b T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$b -o t -- \
    python $R/bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-verify > $O/t$b.log 2>&1
  echo "== blocked $b"; head -4 $O/t$b/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done 2>&1 | tee $O/stats.txt
