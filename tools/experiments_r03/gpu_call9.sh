#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c9; mkdir -p $O
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for v in "A:T360_X=0" "B:T360_DEBUG=64" "C:T360_DEBUG=2" "D:T360_DEBUG=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e T360_PHASES=$O/ph_$n.bin timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-verify > $O/ph_$n.json 2> $O/ph_$n.err
  echo "== $e"; python $R/tools/phase_stats.py $O/ph_$n.bin
done
$R/tools/sweep.sh "T360_X=0" "T360_DEBUG=64" "T360_DEBUG=2" "T360_DEBUG=1" "T360_WGS_PER_XCD=32" "T360_TAIL_PCT=0" "T360_TAIL_PCT=30"
