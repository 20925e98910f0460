#!/bin/bash
# frame clock (pace + wrap start): time and fabric reads vs free running
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c3; mkdir -p $O
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1
CFGS=("T360_PACE=0" "T360_PACE=1" "T360_PACE=80" "T360_PACE=90" "T360_PACE=95" "T360_PACE=100" "T360_PACE=105" "T360_PACE=110" "T360_PACE=120" "T360_PACE=100 T360_TAIL_PCT=0")
$R/tools/sweep.sh "${CFGS[@]}" > $O/sweep.log 2>&1
$R/tools/pmc_rd.sh "${CFGS[@]}" > $O/pmc.log 2>&1
cat $O/sweep.log; grep -E "==|RDREQ_sum" $O/pmc.log
