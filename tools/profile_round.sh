#!/bin/bash
# tools/profile_round.sh <tag> : the profile set committed under profiles/ for one round.
#   1. the plain bench.py line of this build (with --gather-outputs; run last)     -> profiles/<tag>_bench_default.json
#      and the 8-frame batch (BASELINE configs[4] per GPU of an 8-GPU node)        -> profiles/<tag>_bench_8frames.json
#   2. rocprofv3 --kernel-trace --stats of the default run (config 2)              -> profiles/<tag>_kernel_stats.csv
#      of configs 1, 3, 4                                                          -> profiles/<tag>_cfgN_kernel_stats.csv
#      and of the 8-frame batch                                                    -> profiles/<tag>_8frames_kernel_stats.csv
#   3. PMC passes (SQ, TCC memory-side; separate runs, --kernel-trace only)        -> profiles/<tag>_pmc_summary.txt
#      memory-side passes for configs 1, 3, 4 as well                              -> profiles/<tag>_cfgN_pmc_summary.txt
#   4. HBM bytes per launch from the PMC passes (with the library's sha256)        -> profiles/<tag>[_cfgN]_traffic.json
# Run on the GPU box:  gpurun -- "tools/profile_round.sh r04"
set -u
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" --frames 8 --no-cpu-baseline --no-host-abi > "$OUT/bench_8frames.log" 2>&1
grep -h "^{\"metric\"" "$OUT/bench_8frames.log" | tail -1 > "$R/profiles/${TAG}_bench_8frames.json"
trace_cfg() {
  CFG=$1
  SUF=$([ $CFG = 2 ] && echo "" || echo "_cfg$CFG")
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$CFG" -o "$TAG" -- \
      python "$R/bench.py" --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-two-streams --no-native > "$OUT/trace$CFG.log" 2>&1
  cp "$OUT/trace$CFG/${TAG}_kernel_stats.csv" "$R/profiles/${TAG}${SUF}_kernel_stats.csv" 2>/dev/null
  grep -h "^{\"metric\"" "$OUT/trace$CFG.log" | tail -1 > "$R/profiles/${TAG}${SUF}_bench_under_rocprof.json"
}
# (config 2's trace runs at the END, next to the default line: the first processes on a freshly leased box run the same
# launches 3-5 % slower than the ones a few minutes later -- profiles/r05_experiments/README.md call 12 -- and the committed trace average
# must be comparable with the committed line)
for CFG in 1 3 4; do trace_cfg $CFG; done
# config 3 with the low-pass fused into the gather tiles (T360_setFusedLowpass; off by default): which launch takes what
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace3f" -o "$TAG" -- \
    python "$R/bench.py" --config 3 --fused-lowpass --steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-two-streams --no-native > "$OUT/trace3f.log" 2>&1
cp "$OUT/trace3f/${TAG}_kernel_stats.csv" "$R/profiles/${TAG}_cfg3_fused_kernel_stats.csv" 2>/dev/null
grep -h "^{\"metric\"" "$OUT/trace3f.log" | tail -1 > "$R/profiles/${TAG}_cfg3_fused_bench_under_rocprof.json"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace8" -o "$TAG" -- \
    python "$R/bench.py" --frames 8 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-verify --no-two-streams --no-native > "$OUT/trace8.log" 2>&1
cp "$OUT/trace8/${TAG}_kernel_stats.csv" "$R/profiles/${TAG}_8frames_kernel_stats.csv" 2>/dev/null
cd "$R" && PMC_MEM=1 tools/prof_pmc.sh "$OUT/pmc" --no-verify > /dev/null 2>&1
cp "$OUT/pmc/summary.txt" "$R/profiles/${TAG}_pmc_summary.txt"
python tools/make_traffic.py "$OUT/pmc" 2 64 "$R/profiles/${TAG}_traffic.json"
for CFG in 1 3 4; do
  PMC_MEM=only tools/prof_pmc.sh "$OUT/pmc_cfg$CFG" --no-verify --config $CFG > /dev/null 2>&1
  cp "$OUT/pmc_cfg$CFG/summary.txt" "$R/profiles/${TAG}_cfg${CFG}_pmc_summary.txt"
  python tools/make_traffic.py "$OUT/pmc_cfg$CFG" $CFG 64 "$R/profiles/${TAG}_cfg${CFG}_traffic.json"
done
cd /tmp && trace_cfg 2 && cd "$R"
# the default line LAST: it then finds the traffic record of this very build (bench.py compares the library's sha256)
python "$R/bench.py" --gather-outputs > "$OUT/bench_default.log" 2>&1
grep -h "^{\"metric\"" "$OUT/bench_default.log" | tail -1 > "$R/profiles/${TAG}_bench_default.json"
mkdir -p "$R/gpurun_out/profiles" && cp "$R/profiles/${TAG}"* "$R/gpurun_out/profiles/"
head -4 "$R/profiles/${TAG}_kernel_stats.csv" | cut -c1-200
cat "$R/profiles/${TAG}_traffic.json" "$R/profiles/${TAG}_cfg3_traffic.json"
