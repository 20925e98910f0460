#!/bin/bash
# round 4, GPU call 2: what bounds a workgroup's frame period?  Instrumented build, wrong-pixel switches (T360_DEBUG bits:
# 1 no gather, 2 no steady-state DMA, 64 every frame reads frame 0 = L2 hits, 128 no frame barrier, 256 no output stores).
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c2; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 12"
tools/sweep.sh "T360_DEBUG=0" "T360_DEBUG=1" "T360_DEBUG=2" "T360_DEBUG=64" "T360_DEBUG=65" "T360_DEBUG=66" "T360_DEBUG=128" "T360_DEBUG=129" \
  "T360_DEBUG=130" "T360_DEBUG=192" "T360_DEBUG=256" "T360_DEBUG=258" "T360_DEBUG=322" "T360_DEBUG=386" 2>&1 | tee $O/floors.txt
T360_LIB=$R/tools/ab/libT360_slots4.so tools/sweep.sh "T360_DEBUG=0" "T360_DEBUG=1" "T360_DEBUG=65" 2>&1 | tee $O/slots4.txt
