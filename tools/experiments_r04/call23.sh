#!/bin/bash
# round 4, GPU call 23: what does the STAGING alone cost as a function of the address pattern of its LDS-DMA pieces, the
# bytes per frame and the number of frames in flight?  (tools/ubench/ldsdma_pattern.hip)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c23; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsdma tools/ubench/ldsdma_pattern.hip && timeout 300 /tmp/ldsdma > $O/ldsdma_pattern.txt 2>&1
cat $O/ldsdma_pattern.txt
