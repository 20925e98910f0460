#!/bin/bash
# Round 6, call 5: fused path with the x-only R placement and the (kernel, run index, column) lane order; the fused launch on a
# side stream (default) against the same stream, and the two-pass path, interleaved on one box (instrumented build).
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call5; mkdir -p $O
cd $R
python tools/dev/fused_debug.py 2>&1 | grep "frame 0\|frame 24" | grep -v " 0 of " | head
B="python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native"
timeout 300 $B > $O/ship.json 2> $O/ship.err; python - <<PY
import json
d = json.loads(open("$O/ship.json").read().strip().splitlines()[-1]); print("shipped lib: ms/step", d.get("ms_per_step"), "pipelined", (d.get("pipelined") or {}).get("ms_per_step"), "verified", (d.get("verified") or {}).get("max_abs_diff"), d.get("error"))
PY
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for REP in 1 2; do
for ENV in "T360_X=side" "T360_FUSED_SAME_STREAM=1" "T360_NO_FUSED_LOWPASS=1"; do
  env $ENV timeout 300 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
try:
    d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print("$ENV", "ms/step", d.get("ms_per_step"), "pipelined", (d.get("pipelined") or {}).get("ms_per_step"), "verified", (d.get("verified") or {}).get("max_abs_diff"), d.get("error"))
except Exception as e:
    print("$ENV failed", e, open("$O/err.txt").read()[-500:])
PY
done
done
