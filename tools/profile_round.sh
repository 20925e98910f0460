#!/bin/bash
# tools/profile_round.sh <tag> : the profile set committed under profiles/ for one round.
#   1. rocprofv3 --kernel-trace --stats of the default bench.py run  -> profiles/<tag>_kernel_stats.csv
#   2. PMC passes (SQ, TCC memory-side) of the same command           -> profiles/<tag>_pmc_summary.txt
#   3. HBM bytes per launch from the PMC passes                       -> profiles/traffic_latest.json
# Run on the GPU box:  gpurun -- 'tools/profile_round.sh r01'
set -u
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$R/profiles"
FRAMES=${FRAMES:-64}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- \
    python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --frames $FRAMES > "$OUT/trace.log" 2>&1
cp "$OUT/trace/${TAG}_kernel_stats.csv" "$R/profiles/${TAG}_kernel_stats.csv" 2>/dev/null
grep -h "^{\"metric\"" "$OUT/trace.log" | tail -1 > "$R/profiles/${TAG}_bench_under_rocprof.json"
cd "$R" && PMC_MEM=1 tools/prof_pmc.sh "$OUT/pmc" --frames $FRAMES > /dev/null 2>&1
cp "$OUT/pmc/summary.txt" "$R/profiles/${TAG}_pmc_summary.txt"
python tools/make_traffic.py "$OUT/pmc" 2 $FRAMES "$R/profiles/traffic_latest.json"
cp "$R/profiles/"* "$R/gpurun_out/" 2>/dev/null
mkdir -p "$R/gpurun_out/profiles" && cp "$R/profiles/"* "$R/gpurun_out/profiles/"
head -5 "$R/profiles/${TAG}_kernel_stats.csv"
