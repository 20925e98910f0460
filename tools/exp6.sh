#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
tools/sweep.sh "X=1" "T360_BAND=1" "T360_BAND=2" "T360_BAND=3" "T360_BAND=4" "T360_BAND=6" "T360_BAND=8" "T360_BAND=0" "X=2"
export T360_BENCH_ALLOW_INSTRUMENTED=1
tools/pmc_rd.sh "X=1" "T360_BAND=2" "T360_BAND=4" "T360_BAND=8" 2>&1 | grep -E "==|RDREQ_sum"
