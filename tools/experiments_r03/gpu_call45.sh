#!/bin/bash
# 8-frame steps with HBM-resident input vs the same 8 frames every step: per-workgroup timeline (T360_TRACE)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so
mkdir -p gpurun_out/trace8
for mode in ring same; do
  if [ $mode = same ]; then export T360_BENCH_NO_ROTATE=1; fi
  T360_TRACE=$R/gpurun_out/trace8/$mode.bin python bench.py --config 2 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-verify > /dev/null 2>&1
  echo "== $mode =="; python tools/trace_stats.py gpurun_out/trace8/$mode.bin 2>&1 | head -30
done
rm -f gpurun_out/trace8/*.bin
