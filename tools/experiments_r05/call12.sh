#!/bin/bash
# Round 5, call 12: does a process under rocprofv3 --kernel-trace run the hot kernel slower than a plain one on the same box?
# The same short bench.py command alternately plain and under rocprofv3, three rounds.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call12; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-two-streams --no-native --no-verify"
for REP in 1 2 3; do
  timeout 300 $B > $O/plain$REP.json 2> $O/plain$REP.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$REP -o t -- $B > $O/prof$REP.json 2> $O/prof$REP.err
  python - <<PY
import json, csv, glob
a = json.loads(open("$O/plain$REP.json").read().strip().splitlines()[-1])
b = json.loads([l for l in open("$O/prof$REP.json").read().strip().splitlines() if l.startswith('{"metric"')][-1])
k = [r for f in glob.glob("$O/trace$REP/**/t_kernel_stats.csv", recursive=True) for r in csv.reader(open(f)) if "remap_tiled_kernel<4, 76, 8>" in r[0]]
print("round $REP: plain", a["ms_per_step"], a["roofline"]["avg_launch_ms"], "| under rocprofv3", b["ms_per_step"], b["roofline"]["avg_launch_ms"], "trace avg us", [round(float(r[3]) / 1e3, 1) for r in k], "min", [round(float(r[5]) / 1e3, 1) for r in k])
PY
done
rm -rf $O/trace*
