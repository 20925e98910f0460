#!/bin/bash
# Round 6, call 1: (a) tools/ubench/smem_stream (verdict item 1b: the scalar memory path as a staging channel);
# (b) config 3 and config 2 in steps of 8 / 16 / 32 / 64 frames: do the filtered planes of a SHORT step (88 / 177 MB of
# scratch, re-used every step) stay in the 256 MB Infinity Cache between the low-pass launch and the gather?
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call1; mkdir -p $O
cd $R
echo "== smem_stream"; timeout 120 tools/ubench/smem_stream.bin 2>&1 | tee $O/smem_stream.txt
for CFG in 3 2; do
for FR in 64 32 16 8; do
  timeout 300 python bench.py --config $CFG --frames $FR --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native > $O/out.json 2> $O/err.txt
  python - <<PY
import json
try:
    d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    ms = d["ms_per_step"]; p = (d.get("pipelined") or {}).get("ms_per_step")
    print("config $CFG frames $FR: ms/step", ms, "-> per 64 frames", round(ms * 64 / $FR, 4), "| pipelined", p, "-> per 64", (round(p * 64 / $FR, 4) if p else None), "| verified", (d.get("verified") or {}).get("max_abs_diff"))
except Exception as e:
    print("config $CFG frames $FR failed:", e); print(open("$O/err.txt").read()[-600:])
PY
done
done
