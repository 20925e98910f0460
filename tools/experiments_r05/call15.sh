#!/bin/bash
# Round 5, call 15: round 4's blocked filtered planes (config 3: the gather's source is the library's own scratch, written in
# 64x2-pixel blocks per 128-byte line) re-measured WITH the line-read counters (VERDICT round 4, weak point 5): time and
# TCC_EA0_RDREQ of the gather and the low-pass, T360_BLOCKED=0 / 1, instrumented build, one box.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call15; mkdir -p $O
cd $R
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
B="python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --no-two-streams"
for REP in 1 2; do
for BL in 0 1; do
  T360_BLOCKED=$BL timeout 300 $B > $O/out.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/out.json").read().strip().splitlines()[-1])
print("T360_BLOCKED=$BL", "ms/step", d["ms_per_step"], d["repeats_ms_per_step"], "verified", (d.get("verified") or {}).get("max_abs_diff"))
PY
done
done
for BL in 0 1; do
  T360_BLOCKED=$BL PMC_MEM=only tools/prof_pmc.sh $O/pmc$BL --no-verify --config 3 > /dev/null 2>&1
  T360_BLOCKED=$BL timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$BL -o t -- $B --no-verify > $O/trace$BL.log 2>&1
  python - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob("$O/pmc$BL/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        k = "gather" if "remap_tiled" in row["Kernel_Name"] else "lowpass" if "lowpass" in row["Kernel_Name"] else None
        if k: acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    wr = 64 * m.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (m.get("TCC_EA0_WRREQ_sum", 0) - m.get("TCC_EA0_WRREQ_64B_sum", 0))
    print("T360_BLOCKED=$BL", k, "RDREQ %.3f M (128B %.3f M, 64B %.3f M) = %.1f MB read, %.1f MB written; L2 requests %.2f M (reads %.2f M)" % (
        m.get("TCC_EA0_RDREQ_sum", 0) / 1e6, m.get("TCC_EA0_RDREQ_128B_sum", 0) / 1e6, m.get("TCC_EA0_RDREQ_64B_sum", 0) / 1e6, rd / 1e6, wr / 1e6,
        m.get("TCC_REQ_sum", 0) / 1e6, m.get("TCC_READ_sum", 0) / 1e6))
for f in glob.glob("$O/trace$BL/**/t_kernel_stats.csv", recursive=True):
    for r in csv.reader(open(f)):
        if "remap_tiled" in r[0] or "lowpass" in r[0]: print("T360_BLOCKED=$BL", r[0][35:70], "calls", r[1], "avg us %.1f" % (float(r[3]) / 1e3))
PY
done
rm -rf $O/trace0 $O/trace1 $O/pmc0 $O/pmc1
