#!/bin/bash
# round 4, GPU call 26: fabric reads and time of the first one / two / three "generations" of workgroups (slices of every
# XCD's work list, instrumented build, T360_K_HI): do workgroups that START TOGETHER fill fewer lines per tile?
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c26; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
for sl in "0 64" "0 128" "0 192" "64 128" "128 192" "0 0"; do
  set -- $sl
  name=k$1_$2
  for p in mem1 mem2; do
    if [ $p = mem1 ]; then C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; else C="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; fi
    T360_K_LO=$1 T360_K_HI=$2 timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$name/$p -o $p -- \
      python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-abi --no-two-streams --no-verify > $O/$name.$p.log 2>&1 || echo "pass $name $p failed"
  done
done
python - <<PY
import csv, glob, os
from collections import defaultdict
O="$O"
for d in sorted(glob.glob(O+"/k*_*")):
    if not os.path.isdir(d): continue
    acc=defaultdict(list); dur=[]
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "remap_tiled" in r["Kernel_Name"] and r.get("Grid_Size","")=="1048576" or ("remap_tiled" in r["Kernel_Name"] and int(r.get("Grid_Size","0"))>500000):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d+"/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "remap_tiled" in r["Kernel_Name"] and int(r.get("Grid_Size","0"))>500000:
                dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    m={k:sum(v)/len(v) for k,v in acc.items()}
    rd=32*m.get("TCC_EA0_RDREQ_32B_sum",0)+64*m.get("TCC_EA0_RDREQ_64B_sum",0)+128*m.get("TCC_EA0_RDREQ_128B_sum",0)
    wr=64*m.get("TCC_EA0_WRREQ_64B_sum",0)+32*(m.get("TCC_EA0_WRREQ_sum",0)-m.get("TCC_EA0_WRREQ_64B_sum",0))
    dur.sort()
    med=dur[len(dur)//2] if dur else 0
    print("%-10s read %.1f MB write %.1f MB, kernel median %.1f us (n=%d), fabric %.2f TB/s, L2 hit %.1f %%"%(os.path.basename(d), rd/1e6, wr/1e6, med, len(dur), (rd+wr)/med/1e6 if med else 0, 100*m.get("TCC_HIT_sum",0)/max(1,m.get("TCC_HIT_sum",0)+m.get("TCC_MISS_sum",0))))
PY
