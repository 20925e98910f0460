#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
for i in 1 2; do
echo "##### asmread=0"
T360_LIB=$R/tools/ab/libT360_asm0.so $R/tools/sweep.sh "T360_PACE=0" "T360_DEBUG=2" "T360_PACE=100" 2>&1
echo "##### asmread=1"
T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so $R/tools/sweep.sh "T360_PACE=0" "T360_DEBUG=2" "T360_PACE=100" "T360_PACE=95" 2>&1
done
