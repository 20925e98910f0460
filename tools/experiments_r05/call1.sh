#!/bin/bash
# Round 5, call 1: the GPU suite with the pipelined API, the depth sweep of T360_transformFramesPipelined (64- and 8-frame
# steps, config 2 and 3) against the two-handle overlap of round 4, the tail split under pipelining (instrumented build),
# the N1 transfer breakdown, the native driver's multi-worker rehearsal, and a default line.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-verify"
for D in 2 3 4; do
  timeout 300 $B --pipeline-depth $D --two-handles > $O/cfg2_depth$D.json 2> $O/cfg2_depth$D.err
done
for D in 2 3; do
  timeout 300 $B --config 3 --pipeline-depth $D > $O/cfg3_depth$D.json 2> $O/cfg3_depth$D.err
done
# the tail split (12 % of the tiles walk the batch in shorter runs so that a lone launch drains quickly) under pipelining
for TP in 0 6; do
  T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 T360_TAIL_PCT=$TP \
    timeout 300 $B --pipeline-depth 3 > $O/cfg2_depth3_tail$TP.json 2> $O/cfg2_depth3_tail$TP.err
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
          "| 8f", p8.get("ms_per_step"), "8f pipelined", p8.get("pipelined_ms_per_step"), "x", p8.get("speedup_over_1_gpu"), p8.get("pipelined_speedup_over_1_gpu"),
          "| two handles", d.get("two_handles"))
PY
make -C tools/ubench n1_breakdown.bin > /dev/null 2>&1
timeout 120 tools/ubench/n1_breakdown.bin > $O/n1_breakdown.txt 2>&1; cat $O/n1_breakdown.txt
for ARGS in "--workers 1 --frames 64" "--workers 2 --total-frames 64" "--workers 8 --total-frames 64" "--workers 8 --total-frames 64 --pipelined 2" "--workers 1 --frames 8 --pipelined 3"; do
  echo "== examples/t360_multi_gpu $ARGS" >> $O/native_rehearsal.txt
  timeout 120 examples/t360_multi_gpu $ARGS --steps 50 >> $O/native_rehearsal.txt 2>&1
done
tail -30 $O/native_rehearsal.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 1200 $O/bench_default.json
