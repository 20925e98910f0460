#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
T360_LIB=$R/tools/ab/libT360_pt3.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-host-abi --no-two-streams 2>&1 | tail -12 | cut -c1-400
