#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
$R/tools/pmc_rd.sh "T360_X=0" "T360_DEBUG=64" 2>&1 | grep -E "==|RDREQ"
$R/tools/sweep.sh "T360_X=0" "T360_DEBUG=64" "T360_FRAMES_PER_BLOCK=32" "T360_WGS_PER_XCD=48" "T360_WGS_PER_XCD=64" "T360_WGS_PER_XCD=80" "T360_WGS_PER_XCD=128"
