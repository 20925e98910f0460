#!/bin/bash
# Round 6, call 8: config 4 (Lanczos4) after the pole-tile path stopped spilling (the kernel's VGPR count 128 -> 82): shipped
# geometry, and the instrumented build with smaller rings (more workgroups per CU now that registers allow 5-6 waves per SIMD).
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call8; mkdir -p $O
cd $R
B="python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-host-abi --no-native --no-two-streams"
run() { # label env...
  local label=$1; shift
  env "$@" timeout 600 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
try:
    d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print("$label: ms/step", d.get("ms_per_step"), d.get("repeats_ms_per_step"), "verified", (d.get("verified") or {}).get("max_abs_diff"), d.get("roofline", {}).get("kernel", "")[:40], d.get("error"))
except Exception as e:
    print("$label failed", e, open("$O/err.txt").read()[-400:])
PY
}
run "shipped" T360_X=1
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
run "instr ring 38 (4 per CU)" T360_X=1
run "instr ring 31 pieces 12 (5 per CU)" T360_RING_KB=31 T360_MAX_PIECES=12 T360_WAVES=4
run "instr ring 26 pieces 12 (6 per CU)" T360_RING_KB=26 T360_MAX_PIECES=12 T360_WAVES=4
run "instr ring 26 pieces 8 (6 per CU)" T360_RING_KB=26 T360_MAX_PIECES=8 T360_WAVES=4
