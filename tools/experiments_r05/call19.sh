#!/bin/bash
# Round 5, call 19: the blocked filtered planes once more (tools/ab/libT360_blocked.so = the instrumented library with
# profiles/r05_experiments/patches/blocked_scratch_planes_r05.patch): the gather's DMA stream ALONE (T360_DEBUG=1: no gather)
# and the gather ALONE (T360_DEBUG=2: no steady-state DMA) with 17 % fewer line fills -- which side is it that does not move?
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export T360_LIB=$R/tools/ab/libT360_blocked.so T360_BENCH_ALLOW_INSTRUMENTED=1
B="python $R/bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --no-two-streams --no-verify"
# (first: the library really gathers from blocked planes, bit-exact)
for BL in 0 1; do
  T360_BLOCKED=$BL timeout 300 python $R/bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --no-two-streams > $O/v.json 2> $O/v.err
  python -c "
import json; d = json.loads(open('$O/v.json').read().strip().splitlines()[-1]); print('T360_BLOCKED=$BL verified', d['verified']['max_abs_diff'], 'ms/step', d['ms_per_step'])"
done
for REP in 1 2; do
for BL in 0 1; do
for DBG in 0 1 2; do
  T360_BLOCKED=$BL T360_DEBUG=$DBG timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- $B > $O/log.txt 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$O/t/**/t_kernel_stats.csv", recursive=True):
    r = {("gather" if "remap_tiled" in x[0] else "lowpass"): float(x[3]) / 1e3 for x in csv.reader(open(f)) if "remap_tiled" in x[0] or "lowpass" in x[0]}
    print("T360_BLOCKED=$BL T360_DEBUG=$DBG", "gather %.1f us, low-pass %.1f us" % (r.get("gather", 0), r.get("lowpass", 0)))
PY
  rm -rf $O/t
done
done
done
