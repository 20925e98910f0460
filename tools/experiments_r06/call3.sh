#!/bin/bash
# Round 6, call 3: kernel trace of config 3 with the fused path (which launch takes what).
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06_call3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o cfg3 -- python $R/bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-native --no-two-streams --no-verify > $O/bench.json 2> $O/err.txt
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-220'
tail -c 600 $O/bench.json
