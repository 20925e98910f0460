#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c4; mkdir -p $O
cd $R
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for P in 0 1; do
  echo "== PERSIST=$P"
  T360_PERSIST=$P timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-host-abi > $O/b$P.json 2> $O/b$P.err
  tail -c 400 $O/b$P.err; cut -c1-400 $O/b$P.json
done
unset T360_LIB T360_BENCH_ALLOW_INSTRUMENTED
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
