#!/bin/bash
# tail share / tail run length re-tuned on the final kernel (64 frames, config 2)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for rep in 1 2; do
for tf in 8 16 32; do for pct in 6 12 18 25; do
  v=$(T360_TAIL_FRAMES=$tf T360_TAIL_PCT=$pct python bench.py --config 2 --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['strong_cfg5']['ms_per_step'])")
  echo "tail_frames $tf pct $pct: $v"
done; done; done
