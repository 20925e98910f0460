#!/bin/bash
# Round 6, call 7: GPU suite with the fused tests; config 3 two-pass (default) against --fused-lowpass, shipped library.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call7; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -a -E "passed|failed|error" $O/pytest.log | tail -3
for REP in 1 2; do
for ARG in "" "--fused-lowpass"; do
  timeout 300 python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native $ARG > $O/out.json 2> $O/err.txt
  python - <<PY
import json
try:
    d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print("config 3 $ARG: ms/step", d.get("ms_per_step"), "| pipelined", (d.get("pipelined") or {}).get("ms_per_step"), "| verified", (d.get("verified") or {}).get("max_abs_diff"), "|", d.get("roofline", {}).get("kernel"), d.get("error"))
except Exception as e:
    print("failed:", e); print(open("$O/err.txt").read()[-800:])
PY
done
done
