#!/bin/bash
# round 4, GPU call 24: HBM delivery rate for a plane read as row fragments of S bytes (tools/ubench/hbm_fragments.hip)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c24; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbmfrag tools/ubench/hbm_fragments.hip && timeout 300 /tmp/hbmfrag > $O/hbm_fragments.txt 2>&1
cat $O/hbm_fragments.txt
