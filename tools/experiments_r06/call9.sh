#!/bin/bash
# Round 6, call 9: the hot kernel at ONE workgroup per CU (T360_LDS_PAD: more LDS per workgroup than the ring needs): if the
# DMA-bound launch keeps its pace at half the residency, the other half of every CU could run the next batch's low-pass
# (integer-VALU-bound) at the same time.  Instrumented build; config 2 and config 3, plain and pipelined.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call9; mkdir -p $O
cd $R
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for CFG in 2 3; do
for PAD in 0 12000; do
for DEPTH in 2 3; do
  T360_LDS_PAD=$PAD timeout 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --pipeline-depth $DEPTH > $O/out.json 2> $O/err.txt
  python - <<PY
import json
try:
    d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print("config $CFG lds_pad $PAD depth $DEPTH: ms/step", d.get("ms_per_step"), "| pipelined", (d.get("pipelined") or {}).get("ms_per_step"), "| verified", (d.get("verified") or {}).get("max_abs_diff"), d.get("error"))
except Exception as e:
    print("failed:", e); print(open("$O/err.txt").read()[-500:])
PY
done
done
done
