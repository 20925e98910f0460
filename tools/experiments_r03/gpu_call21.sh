#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
CFGS=("T360_SYNC=0" "T360_SYNC=3" "T360_SYNC=4 T360_SYNC_WINDOW=10" "T360_SYNC=6 T360_SYNC_WINDOW=12" "T360_SYNC=8 T360_SYNC_WINDOW=16" "T360_SYNC=100" "T360_SYNC=0")
$R/tools/sweep.sh "${CFGS[@]}" 2>&1
$R/tools/pmc_rd.sh "${CFGS[@]}" 2>&1 | grep -E "==|RDREQ_sum"
