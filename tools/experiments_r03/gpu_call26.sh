#!/bin/bash
# half-line fetch probe: timing + fabric read request sizes per kernel
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/half_line; mkdir -p $O
timeout 120 $R/tools/ubench/half_line.bin | tee $O/timing.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc -o hl -- $R/tools/ubench/half_line.bin > $O/pmc.log 2>&1
python - <<P
import csv,glob,collections
f=glob.glob("$O/pmc/**/hl_counter_collection.csv",recursive=True)
acc=collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"][:60]; acc.setdefault(k,collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)/1e6,3) for c,x in v.items()}, "M per launch")
P
