#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
echo "#### P=24 LDS 80"
T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so $R/tools/sweep.sh "T360_X=0" "T360_DEBUG=1" "T360_MAX_PIECES=22"
echo "#### P=22 LDS 76"
T360_LIB=$R/tools/ab/libT360_p22.so $R/tools/sweep.sh "T360_X=0" "T360_DEBUG=1"
