#!/bin/bash
# round 4, GPU call 17: where do scatter tiles lose?  floors (no gather / no DMA / no stores) and LDS / L2 / write counters
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c17; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 12"
tools/sweep.sh "T360_SCATTER=2" "T360_SCATTER=2 T360_DEBUG=1" "T360_SCATTER=2 T360_DEBUG=2" "T360_SCATTER=2 T360_DEBUG=256" "T360_SCATTER=2 T360_DEBUG=258" 2>&1 | tee $O/floors.txt
cd /tmp && export TMPDIR=/tmp
for sc in 0 2; do
  for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_READ_sum"; do
    T360_SCATTER=$sc T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_${sc}_$(echo $pass | cut -c1-6) -o p -- \
      python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-abi --no-two-streams --no-verify > /dev/null 2>&1
  done
  python - <<PY
import csv,glob
acc={}
for p in glob.glob("$O/pmc_${sc}_*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "remap_tiled" in r["Kernel_Name"] and "76, 8" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("scatter $sc:", {k: round(sum(v)/len(v)/1e6,2) for k,v in sorted(acc.items()) if len(v)>20})
PY
done 2>&1 | tee $O/pmc.txt
