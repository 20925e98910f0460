#!/bin/bash
# Round 5, call 18: K <= 4 ring slots for the 8-wave kernel in the SHIPPED configuration (A/B builds without the instrumented
# switches, tools/ab_build.sh with NOINSTR=1), configs 2 and 3, verified, three interleaved rounds.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call18; mkdir -p $O
cd $R
for CFG in 2 3; do
B="python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --no-two-streams"
for REP in 1 2 3; do
for V in k3 k4; do
  T360_LIB=$R/tools/ab/libT360_$V.so timeout 300 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
p8 = (d.get("strong_cfg5") or {}).get("projected_8_gpus") or {}
print("cfg $CFG $V", "ms/step", d["ms_per_step"], d["repeats_ms_per_step"], "8f", p8.get("ms_per_step"), "verified", (d.get("verified") or {}).get("max_abs_diff"))
PY
done
done
done
