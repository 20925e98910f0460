#!/bin/bash
# Round 5, call 16: what the per-frame workgroup barrier costs WITH the DMA running (round 4 measured it without: 0.1647 vs
# 0.1956): config 2, instrumented build, T360_DEBUG bit 7 = no barrier (pixels are wrong, times are not), bit 0 = no gather,
# bit 1 = no steady-state DMA, bit 6 = every frame reads frame 0.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call16; mkdir -p $O
cd $R
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --no-two-streams --no-verify"
for REP in 1 2; do
for DBG in 0 128 2 130 1 129 64 192; do
  T360_DEBUG=$DBG timeout 300 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
print("T360_DEBUG=$DBG", "ms/step", d["ms_per_step"], d["repeats_ms_per_step"])
PY
done
done
