#!/bin/bash
# Round 5, call 13: wave priority (s_setprio 3) while a wave issues its DMA / its deferred store + DMA (instrumented build,
# T360_DEBUG bits 9 / 10; right pixels): do the requests of a memory-bound kernel leave earlier?
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call13; mkdir -p $O
cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-two-streams --no-native"
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for REP in 1 2 3; do
for ENV in "T360_DEBUG=0" "T360_DEBUG=512" "T360_DEBUG=1024"; do
  env $ENV timeout 300 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
p8 = d["strong_cfg5"]["projected_8_gpus"]
print("$ENV", "ms/step", d["ms_per_step"], d["repeats_ms_per_step"], "8f", p8["ms_per_step"], "verified", (d.get("verified") or {}).get("max_abs_diff"))
PY
done
done
