#!/bin/bash
# 8 frames per step: tail runs shorter than the batch (T360_TAIL_FRAMES) for a share of the tiles (T360_TAIL_PCT)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for f in 8 12 16; do
for tf in 16 4 2; do for pct in 12 30 60; do
  v=$(T360_TAIL_FRAMES=$tf T360_TAIL_PCT=$pct python bench.py --config 2 --frames $f --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['strong_cfg5']['ms_per_step'])")
  echo "frames $f tail_frames $tf pct $pct: $v"
  [ $tf = 16 ] && break
done; done; done
