#!/bin/bash
# workgroups of the 4-wave plan walk several work items each (T360_MULTI_ITEM 0 off, 1 small plan, 2 Lanczos4 too)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
for rep in 1 2; do
for f in 1 4 8 12 16 23; do
 line="frames $f:"
 for mi in 0 1; do
  v=$(T360_MULTI_ITEM=$mi python bench.py --config 2 --frames $f --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['strong_cfg5']['ms_per_step'], d['strong_cfg5'].get('ms_per_step_same_input_every_step'), 'verified', d['verified']['max_abs_diff'])")
  line="$line | multi_item $mi: $v"
 done
 echo "$line"
done; done
for mi in 0 2; do
  v=$(T360_MULTI_ITEM=$mi python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'verified', d['verified']['max_abs_diff'])")
  echo "config 4 multi_item $mi: $v"
done
