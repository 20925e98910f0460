#!/bin/bash
# short steps with HBM-resident input (the bench's input ring): ring depth, plan choice, tail share
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1
run() { # label, env..., -- args
  local label=$1; shift
  local v=$(env "$@" python bench.py $ARGS --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'][:27])")
  echo "$ARGS | $label: $v"
}
for f in 8 16 24 32; do
 ARGS="--config 2 --frames $f"
 run base T360_LIB=$R/tools/ab/libT360_base.so
 run slots4 T360_LIB=$R/tools/ab/libT360_slots4.so
 run "8-wave plan" T360_LIB=$R/tools/ab/libT360_base.so T360_SMALL_BATCH=1
 run "4-wave plan" T360_LIB=$R/tools/ab/libT360_base.so T360_SMALL_BATCH=64
 run "tail 30%" T360_LIB=$R/tools/ab/libT360_base.so T360_TAIL_PCT=30
 run "tail 0%" T360_LIB=$R/tools/ab/libT360_base.so T360_TAIL_PCT=0
done
