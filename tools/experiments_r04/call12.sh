#!/bin/bash
# round 4, GPU call 12: host-pointer ABI (3 synchronous calls per frame): polling the stream instead of sleeping on it
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c12; mkdir -p $O
cd $R
for rep in 1 2; do
for lib in transform360_amd/lib/libTransform360_instr.so tools/ab/libT360_prestore.so; do
  T360_LIB=$R/$lib T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib', d['host_abi']['frames_per_s'], d['host_abi']['ms_per_frame'], d['host_abi']['pcie_GBps_in_plus_out'])"
done; done 2>&1 | tee $O/host_abi.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host" 2>&1 | tail -2
