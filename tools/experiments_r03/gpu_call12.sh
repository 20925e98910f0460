#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c12; mkdir -p $O
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for v in "A:T360_X=0:64" "B:T360_DEBUG=64:64" "C:T360_X=0:8"; do
  n=${v%%:*}; r=${v#*:}; e=${r%%:*}; f=${r#*:}
  env $e T360_PHASES=$O/ph_$n.bin timeout 300 python $R/bench.py --frames $f --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-verify > $O/ph_$n.json 2> $O/ph_$n.err
  echo "== $e frames $f"; python $R/tools/phase_stats.py $O/ph_$n.bin | grep -v pieces
done
$R/tools/pmc_rd.sh "T360_X=0" 2>&1 | grep -E "==|RDREQ_sum"
