#!/bin/bash
# Round 6, call 2: first run of the fused low-pass + gather (remap_fused_kernel): GPU suite, then config 3 and config 2 lines.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call2; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -a -E "passed|failed|error" $O/pytest.log | tail -3
for CFG in 3 2; do
  timeout 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native > $O/out$CFG.json 2> $O/err$CFG.txt
  python - <<PY
import json
try:
    d = json.loads(open("$O/out$CFG.json").read().strip().splitlines()[-1])
    print("config $CFG: ms/step", d.get("ms_per_step"), d.get("repeats_ms_per_step"), "| pipelined", (d.get("pipelined") or {}).get("ms_per_step"), "| verified", d.get("verified"), "|", d.get("roofline", {}).get("kernel"), d.get("error"))
except Exception as e:
    print("config $CFG failed:", e); print(open("$O/err$CFG.txt").read()[-1500:])
PY
done
