#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
for i in 1 2 3; do
echo "## slots3"; T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so $R/tools/sweep.sh "T360_X=0" 2>&1 | tail -1
echo "## slots4"; T360_LIB=$R/tools/ab/libT360_slots4.so $R/tools/sweep.sh "T360_X=0" 2>&1 | tail -1
done
T360_LIB=$R/tools/ab/libT360_slots4.so $R/tools/pmc_rd.sh "T360_X=0" 2>&1 | grep -E "RDREQ_sum"
BENCH_EXTRA="--frames 32" T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so $R/tools/sweep.sh "T360_X=0" 2>&1 | tail -1
BENCH_EXTRA="--frames 32" T360_LIB=$R/tools/ab/libT360_slots4.so $R/tools/sweep.sh "T360_X=0" 2>&1 | tail -1
BENCH_EXTRA="--frames 8" T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so $R/tools/sweep.sh "T360_X=0" 2>&1 | tail -1
BENCH_EXTRA="--frames 8" T360_LIB=$R/tools/ab/libT360_slots4.so $R/tools/sweep.sh "T360_X=0" 2>&1 | tail -1
