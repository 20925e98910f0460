#!/bin/bash
# second long fuzz soak of the final build (other seeds, larger counts)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/soak; mkdir -p $O
cd $R
{
sha256sum transform360_amd/lib/libTransform360.so | cut -c1-16
for m in plane batch plane4 tiny; do
  case $m in plane) n=15000;; batch) n=3000;; plane4) n=1500;; tiny) n=15000;; esac
  echo "mode $m, $n seeds from 40000:"
  timeout 900 python tests/soak/fuzz_soak.py 40000 $n $m > $O/long2_$m.log 2>&1; echo "exit $?"; grep -a -o "seeds [0-9.]*: [0-9]* mismatches" $O/long2_$m.log | tail -2
done
for c in 2 3; do timeout 300 python tools/soak.py $c 1000 2>&1 | grep -a "iterations"; done
} 2>&1 | tee $O/soak_long2.txt
