#!/bin/bash
# Round 5, call 8: config 1 (nearest) with 64x16 / 128x16 tiles chosen by the lines under their row fragments
# (PlanOptions::cost_lines; nearest maps took 32x32 tiles only): 4-wave and 8-wave plans, instrumented build; second run: 128x8
# strips and 256x8 tiles on top.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call8; mkdir -p $O
cd $R
B="python bench.py --config 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-two-streams"
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for REP in 1 2; do
for ENV in "T360_COST_LINES=1" "T360_COST_LINES=1 T360_STRIPS=100" "T360_COST_LINES=1 T360_STRIPS=120" "T360_COST_LINES=1 T360_SMALL_BATCH=0 T360_WIDE256=100" "T360_COST_LINES=1 T360_SMALL_BATCH=0 T360_WIDE256=120" "T360_COST_LINES=1 T360_MAX_PIECES=8"; do
  env $ENV timeout 300 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
print("$ENV", "ms/step", d["ms_per_step"], d["roofline"]["kernel"][:32], "verified", (d.get("verified") or {}).get("max_abs_diff"), str(d["gather_plan"][0])[:60])
PY
done
done
