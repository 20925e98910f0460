#!/bin/bash
# Round 6, call 4: PMC counters of config 3 with the fused path.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call4; mkdir -p $O
cd $R
PMC_MEM=0 bash tools/prof_pmc.sh $O/pmc --config 3 > $O/pmc.log 2>&1
grep -A30 "remap_fused_kernel" $O/pmc/summary.txt | head -45
