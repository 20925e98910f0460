#!/bin/bash
# round 6: determinism and fuzz soaks of the final build (round 5's modes, new seeds) + the fused low-pass soak -> profiles/r06_soak.txt
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/soak_r06; mkdir -p $O
cd $R
{
echo "library sha256[:16] $(sha256sum transform360_amd/lib/libTransform360.so | cut -c1-16)"
for cfg in 2 3 1; do timeout 300 python tools/soak.py $cfg 300 2>&1 | tail -1; done
timeout 300 python tools/soak.py 4 40 2>&1 | tail -1
for m in plane batch plane4 tiny; do
  case $m in plane) n=5000;; batch) n=1200;; plane4) n=600;; tiny) n=5000;; esac
  echo "mode $m, $n seeds from 200000:"
  timeout 900 python tests/soak/fuzz_soak.py 200000 $n $m > $O/long_$m.log 2>&1; echo "exit $?"; grep -a "seeds\|mismatch\|differ" $O/long_$m.log | tail -3
done
timeout 900 python tests/soak/fused_soak.py 1000 1000 2>&1 | grep -a "fused soak"
} 2>&1 | tee $O/soak.txt
