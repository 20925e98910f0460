#!/bin/bash
# round 4, GPU call 1: LDS read forms, MALL vs HBM streaming rates, the GPU suite on the round's first tree, and the
# first A/B set on the hot kernel (DMA-first prologue, 4-mod-8 ds_read_b64, 256x8 tiles).
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c1; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_b64 tools/ubench/lds_b64_align4.hip && timeout 120 /tmp/lds_b64 > $O/lds_b64.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe tools/ubench/mall_probe.hip && timeout 300 /tmp/mall_probe > $O/mall_probe.txt 2>&1
cat $O/lds_b64.txt $O/mall_probe.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
export BENCH_EXTRA="--steps 20"
tools/sweep.sh "T360_X=base" "T360_WIDE256=100 T360_COST_LINES=1" "T360_WIDE256=115 T360_COST_LINES=1" "T360_WIDE256=1000" 2>&1 | tee $O/sweep_instr.txt
for v in dmafirst b64m both; do
  T360_LIB=$R/tools/ab/libT360_$v.so tools/sweep.sh "T360_X=$v" 2>&1 | tee -a $O/sweep_variants.txt
done
# verification of the b64m build against the oracle (bench.py verifies frames unless --no-verify)
T360_LIB=$R/tools/ab/libT360_both.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-host-abi 2>&1 | tail -1 | cut -c1-1500 | tee $O/both_verified.json
T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so tools/sweep.sh "T360_X=base_again" 2>&1 | tee -a $O/sweep_instr.txt
