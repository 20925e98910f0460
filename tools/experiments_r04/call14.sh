#!/bin/bash
# round 4, GPU call 14: pre-touch one frame ahead; fabric reads and L2 hits of the pre-touch builds (PMC)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c14; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_X=base" 2>&1 | tee $O/sweep.txt
T360_LIB=$R/tools/ab/libT360_pt1.so tools/sweep.sh "T360_X=pt1" 2>&1 | tee -a $O/sweep.txt
cd /tmp && export TMPDIR=/tmp
for v in base pt1 pt2; do
  LIB=$R/tools/ab/libT360_$v.so; [ $v = base ] && LIB=$R/transform360_amd/lib/libTransform360_instr.so
  T360_LIB=$LIB T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_$v -o p -- \
    python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-abi --no-two-streams --no-verify > $O/pmc_$v.log 2>&1
  python - <<PY
import csv,glob
acc={}
for p in glob.glob("$O/pmc_$v/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "remap_tiled" in r["Kernel_Name"] and r.get("Grid_Size")=="1048576":
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("$v", {k: round(sum(v)/len(v)/1e6,3) for k,v in acc.items()}, "M per launch")
PY
done 2>&1 | tee -a $O/sweep.txt
