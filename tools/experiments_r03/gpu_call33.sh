#!/bin/bash
# low-pass: staging index math / column-pass taps from SGPRs, separately and together (config 3, per-kernel times from rocprofv3)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
export T360_BENCH_ALLOW_INSTRUMENTED=1
for rep in 1 2 3; do
for v in base lpmath lpsreg lpboth2; do
T360_LIB=$R/tools/ab/libT360_$v.so python bench.py --config 3 --no-cpu-baseline --no-host-abi --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v |', d['ms_per_step'])"
done; done
cd /tmp && export TMPDIR=/tmp
for v in base lpboth2; do
T360_LIB=$R/tools/ab/libT360_$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lpk_$v -o k -- python $R/bench.py --config 3 --no-cpu-baseline --no-host-abi --no-verify > /dev/null 2>&1
echo $v; grep -h "lowpass\|remap_tiled" $R/gpurun_out/lpk_$v/*kernel_stats.csv $R/gpurun_out/lpk_$v/*/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4,6,7 | cut -c1-150
done
