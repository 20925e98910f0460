#!/bin/bash
# frame-group-major order: time and fabric reads vs the default
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c2; mkdir -p $O
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
CFGS=("T360_FG_MAJOR=0" "T360_FG_MAJOR=1 T360_FRAMES_PER_BLOCK=32" "T360_FG_MAJOR=1 T360_FRAMES_PER_BLOCK=16" "T360_FG_MAJOR=1 T360_FRAMES_PER_BLOCK=8" "T360_FG_MAJOR=1 T360_FRAMES_PER_BLOCK=64" "T360_FG_MAJOR=0 T360_FRAMES_PER_BLOCK=16 T360_TAIL_PCT=0" "T360_FG_MAJOR=1 T360_FRAMES_PER_BLOCK=16 T360_WAVES=4 T360_RING_KB=38 T360_MAX_PIECES=12" "T360_FG_MAJOR=1 T360_FRAMES_PER_BLOCK=8 T360_WAVES=4 T360_RING_KB=38 T360_MAX_PIECES=12")
$R/tools/sweep.sh "${CFGS[@]}" > $O/sweep.log 2>&1
$R/tools/pmc_rd.sh "${CFGS[@]}" > $O/pmc.log 2>&1
cat $O/sweep.log; grep -E "==|RDREQ_128B|RDREQ_sum" $O/pmc.log
