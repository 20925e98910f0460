#!/bin/bash
# Round 5, call 17: ring depth again, now that the memory side is known to be latency-bound per workgroup (call 16): K <= 3 slots
# (two frames' DMA in flight) against K <= 4 (three), A/B builds of the instrumented library (tools/ab_build.sh), config 2.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call17; mkdir -p $O
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --no-two-streams"
for REP in 1 2 3; do
for V in s3 s4; do
  for DBG in 0 1; do
  T360_LIB=$R/tools/ab/libT360_$V.so T360_DEBUG=$DBG timeout 300 $B $([ $DBG = 1 ] && echo --no-verify) > $O/out.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
p8 = (d.get("strong_cfg5") or {}).get("projected_8_gpus") or {}
print("$V T360_DEBUG=$DBG", "ms/step", d["ms_per_step"], d["repeats_ms_per_step"], "8f", p8.get("ms_per_step"), "verified", (d.get("verified") or {}).get("max_abs_diff"))
PY
  done
done
done
