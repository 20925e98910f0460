#!/bin/bash
# where the 4-wave plan hands over to the 8-wave plan, with balanced tail runs (T360_SMALL_BATCH: batches below it use the 4-wave plan)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for f in 24 28 32 36 40 48; do
 line="frames $f:"
 for sb in 1 64; do
  v=$(T360_SMALL_BATCH=$sb python bench.py --config 2 --frames $f --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['strong_cfg5']['ms_per_step'], d['roofline']['kernel'][:27])")
  line="$line | small_batch $sb: $v"
 done
 echo "$line"
done
