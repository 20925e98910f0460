#!/bin/bash
# Round 5, call 11: config 3 with the low-pass of Y, U and V as ONE launch (lowpass_q8w_multi_kernel) against three launches on
# three streams (instrumented build, T360_NO_MERGED_LOWPASS), interleaved on one box; the GPU suite on the new build.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call11; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -a -o -E "[0-9]+ passed[^\n]{0,60}|[0-9]+ failed[^\n]{0,60}" $O/pytest.log | tail -1
B="python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native"
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for REP in 1 2 3; do
for ENV in "T360_X=merged" "T360_NO_MERGED_LOWPASS=1"; do
  env $ENV timeout 300 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
print("$ENV", "ms/step", d["ms_per_step"], d["repeats_ms_per_step"], "pipelined", (d.get("pipelined") or {}).get("ms_per_step"), "verified", (d.get("verified") or {}).get("max_abs_diff"))
PY
done
done
