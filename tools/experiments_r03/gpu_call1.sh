#!/bin/bash
# round-3 first GPU call: baseline of this box, 8-frame batch, L2 probe, workgroup trace of the 64-frame launch
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c1
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-host-abi > $O/bench64.json 2> $O/bench64.err
python bench.py --no-cpu-baseline --no-host-abi --frames 8 --no-verify > $O/bench8.json 2> $O/bench8.err
# trace of one 64-frame launch (instrumented build)
T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 T360_TRACE=$O/trace64.bin \
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-abi --no-verify > $O/trace64.json 2> $O/trace64.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2probe -o l2 -- $R/tools/ubench/l2_probe.bin > $O/l2probe.log 2>&1
python $R/tools/ubench/l2_probe_report.py $O/l2probe > $O/l2probe_report.txt 2>&1
cat $O/l2probe_report.txt
tail -c 600 $O/bench64.json; echo; tail -c 300 $O/bench8.json
