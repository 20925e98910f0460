#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1
echo "#### old"; T360_LIB=$R/tools/ab/libT360_old.so $R/tools/sweep.sh "T360_X=0" "T360_DEBUG=1" "T360_DEBUG=2"
echo "#### new"; T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so $R/tools/sweep.sh "T360_X=0" "T360_DEBUG=1" "T360_DEBUG=2"
echo "#### new static items, one per workgroup"; T360_LIB=$R/tools/ab/libT360_static.so $R/tools/sweep.sh "T360_WGS_PER_XCD=256" "T360_WGS_PER_XCD=256 T360_DEBUG=1" "T360_WGS_PER_XCD=256 T360_DEBUG=2" "T360_WGS_PER_XCD=64"
echo "#### old again"; T360_LIB=$R/tools/ab/libT360_old.so $R/tools/sweep.sh "T360_X=0"
