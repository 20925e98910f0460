#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
CFGS=("T360_PACE=0" "T360_PACE=100" "T360_PACE=100 T360_PACE_LEAD=1" "T360_PACE=100 T360_PACE_LEAD=2" "T360_PACE=95 T360_PACE_LEAD=2" "T360_PACE=90 T360_PACE_LEAD=2" "T360_PACE=90 T360_PACE_LEAD=4" "T360_PACE=85 T360_PACE_LEAD=3" "T360_PACE=105 T360_PACE_LEAD=2" "T360_PACE=0")
$R/tools/sweep.sh "${CFGS[@]}" 2>&1
$R/tools/pmc_rd.sh "${CFGS[@]}" 2>&1 | grep -E "==|RDREQ_sum"
