#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
$R/tools/sweep.sh "T360_X=0" "T360_WGS_PER_XCD=256" "T360_WGS_PER_XCD=128" "T360_WGS_PER_XCD=96" "T360_WGS_PER_XCD=256 T360_DEBUG=1" "T360_WGS_PER_XCD=256 T360_DEBUG=2"
$R/tools/pmc_rd.sh "T360_WGS_PER_XCD=256" 2>&1 | grep -E "==|RDREQ_sum"
