#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c5; mkdir -p $O
export T360_BENCH_ALLOW_INSTRUMENTED=1
echo "##### slots3 (default instr)"
T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so $R/tools/sweep.sh "T360_PACE=0" "T360_DEBUG=1" "T360_DEBUG=2" "T360_PACE=100" 2>&1
echo "##### slots4"
T360_LIB=$R/tools/ab/libT360_slots4.so $R/tools/sweep.sh "T360_PACE=0" "T360_DEBUG=1" "T360_DEBUG=2" "T360_PACE=100" "T360_PACE=90" "T360_PACE=80" 2>&1
T360_LIB=$R/tools/ab/libT360_slots4.so $R/tools/pmc_rd.sh "T360_PACE=0" "T360_PACE=90" 2>&1 | grep -E "==|RDREQ_sum"
