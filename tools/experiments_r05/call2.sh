#!/bin/bash
# Round 5, call 2: the pipelined legs with the benchmark's host overhead out of the timed loop (bench.py fast_steps),
# depth 2 / 3 / 4, configs 2, 3, 1; the native driver with a start gate: 8 workers x 8 frames on one GPU against one
# worker x 64, and one worker x 8 frames at depth 0 .. 4 (no Python in the loop at all).
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call2; mkdir -p $O
cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-verify"
for D in 2 3 4; do
  timeout 300 $B --pipeline-depth $D > $O/cfg2_depth$D.json 2> $O/cfg2_depth$D.err
done
timeout 300 $B --config 3 --pipeline-depth 3 > $O/cfg3_depth3.json 2> $O/cfg3_depth3.err
timeout 300 $B --config 1 --pipeline-depth 3 > $O/cfg1_depth3.json 2> $O/cfg1_depth3.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/cfg*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "no line", e); continue
    s = d.get("strong_cfg5") or {}
    p8 = s.get("projected_8_gpus") or {}
    print(os.path.basename(f), "ms/step", d["ms_per_step"], "pipelined", (d.get("pipelined") or {}).get("ms_per_step"),
          "| 8f", p8.get("ms_per_step"), "8f pipelined", p8.get("pipelined_ms_per_step"), "x", p8.get("speedup_over_1_gpu"), p8.get("pipelined_speedup_over_1_gpu"))
PY
for ARGS in "--workers 1 --frames 64" "--workers 1 --frames 64 --pipelined 3" "--workers 8 --total-frames 64" "--workers 8 --total-frames 64 --pipelined 2" "--workers 4 --total-frames 64" "--workers 2 --total-frames 64 --pipelined 2" \
            "--workers 1 --frames 8" "--workers 1 --frames 8 --pipelined 1" "--workers 1 --frames 8 --pipelined 2" "--workers 1 --frames 8 --pipelined 3" "--workers 1 --frames 8 --pipelined 4" \
            "--workers 1 --frames 16 --pipelined 3" "--workers 1 --frames 32 --pipelined 3"; do
  echo "== examples/t360_multi_gpu $ARGS" >> $O/native.txt
  timeout 120 examples/t360_multi_gpu $ARGS --steps 200 2>&1 | tail -1 >> $O/native.txt
done
cat $O/native.txt
