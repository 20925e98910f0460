#!/bin/bash
# long fuzz soak of the final build: random configurations far beyond the suite's seeds
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/soak; mkdir -p $O
cd $R
{
sha256sum transform360_amd/lib/libTransform360.so | cut -c1-16
for m in plane batch plane4 tiny; do
  case $m in plane) n=6000;; batch) n=1500;; plane4) n=800;; tiny) n=6000;; esac
  echo "mode $m, $n seeds from 20000:"
  timeout 500 python tests/soak/fuzz_soak.py 20000 $n $m > $O/long_$m.log 2>&1; echo "exit $?"; grep -a "seeds\|mismatch\|differ" $O/long_$m.log | tail -4
done
} 2>&1 | tee $O/soak_long.txt
