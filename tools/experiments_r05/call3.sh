#!/bin/bash
# Round 5, call 3: short steps from HBM without Python in the loop (examples/t360_multi_gpu, input ring of 320 MB):
# depth 0 .. 4 with the idle-stream shortcut; under pipelining, does the 8-wave plan (less halo traffic, costlier start-up)
# or an unsplit tail beat the defaults (4-wave plan below 24 frames, last 12 % of the tiles in two runs)?
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call3; mkdir -p $O
cd $R
run() { echo "== $*" >> $O/native.txt; timeout 120 "$@" 2>&1 | tail -1 >> $O/native.txt; }
for D in 0 1 2 3 4; do
  P=$([ $D = 0 ] && echo "" || echo "--pipelined $D")
  run examples/t360_multi_gpu --workers 1 --frames 8 --steps 400 $P
done
for D in 0 2 3 4; do
  P=$([ $D = 0 ] && echo "" || echo "--pipelined $D")
  run examples/t360_multi_gpu --workers 1 --frames 64 --steps 100 $P
done
run examples/t360_multi_gpu --workers 1 --frames 8 --steps 400 --ring-mb 0
run examples/t360_multi_gpu --workers 1 --frames 8 --steps 400 --ring-mb 0 --pipelined 3
for D in 0 3 4; do
  P=$([ $D = 0 ] && echo "" || echo "--pipelined $D")
  for ENV in "T360_X=0" "T360_SMALL_BATCH=0" "T360_TAIL_PCT=0" "T360_SMALL_BATCH=0 T360_TAIL_PCT=0" "T360_TAIL_PCT=25" "T360_SMALL_BATCH=0 T360_TAIL_FRAMES=4"; do
    run env $ENV examples/t360_multi_gpu_instr --workers 1 --frames 8 --steps 400 $P
  done
done
for ENV in "T360_X=0" "T360_TAIL_PCT=0" "T360_TAIL_PCT=6"; do
  run env $ENV examples/t360_multi_gpu_instr --workers 1 --frames 64 --steps 100 --pipelined 3
done
cat $O/native.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-verify"
for D in 2 3 4; do
  timeout 300 $B --pipeline-depth $D > $O/cfg2_depth$D.json 2> $O/cfg2_depth$D.err
done
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
