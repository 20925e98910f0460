#!/bin/bash
# Round 6, call 6: where does the fused kernel's time go?  Kernel traces of config 3 (instrumented build) with parts of the
# fused kernel switched off (T360_DEBUG bits: 1 no gather, 2 no steady DMA, 2048 no filter, 4096 no blurred writes, 8192 no
# second barrier); tools/ubench/valu_rate.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call6; mkdir -p $O
cd $R
tools/ubench/valu_rate.bin 2>&1 | tail -12
export T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 T360_FUSED_SAME_STREAM=1
cd /tmp && export TMPDIR=/tmp
for DBG in 0 1 2 2048 4096 6144 8192 6145 6147 2049; do
  rm -rf $O/trace
  T360_DEBUG=$DBG timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-host-abi --no-native --no-two-streams --no-verify > $O/bench.json 2> $O/err.txt
  echo "T360_DEBUG=$DBG: $(grep remap_fused_kernel $O/trace/t_kernel_stats.csv | awk -F, '{printf "fused %.1f us", $4/1000}') $(grep 'remap_tiled_kernel' $O/trace/t_kernel_stats.csv | awk -F, '{printf "tiled %.1f us", $4/1000}') $(grep 'lowpass' $O/trace/t_kernel_stats.csv | awk -F, '{printf "lowpass %.1f us", $4/1000}')"
done
