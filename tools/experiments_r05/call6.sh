#!/bin/bash
# Round 5, calls 6-7: bench.py on a stream of its own (not the legacy NULL stream), input ring of three 64-frame batches; call 7: the
# pipelined legs issue their K calls through T360_transformFramesPipelinedMany (no interpreter in the loop)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call7; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -1 $O/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi"
for D in 2 2; do
  timeout 300 $B --pipeline-depth $D > $O/cfg2_depth$D.json 2> $O/cfg2_depth$D.err
  python - <<PY
import json
d = json.loads(open("$O/cfg2_depth$D.json").read().strip().splitlines()[-1])
s = d.get("strong_cfg5") or {}
p8 = s.get("projected_8_gpus") or {}
print("depth $D ms/step", d["ms_per_step"], d["repeats_ms_per_step"], "pipelined", (d.get("pipelined") or {}).get("ms_per_step"),
      "| 8f", p8.get("ms_per_step"), "8f pipelined", p8.get("pipelined_ms_per_step"), "x", p8.get("speedup_over_1_gpu"), p8.get("pipelined_speedup_over_1_gpu"), "verified", (d.get("verified") or {}).get("max_abs_diff"))
PY
done
run() { echo "== $*"; timeout 120 "$@" 2>&1 | tail -1; }
run examples/t360_multi_gpu --workers 1 --frames 8 --steps 400 --ring-mb 1440
run examples/t360_multi_gpu --workers 1 --frames 8 --steps 400 --ring-mb 1440 --pipelined 2
run examples/t360_multi_gpu --workers 1 --frames 64 --steps 100 --ring-mb 1440
