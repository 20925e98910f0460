#!/bin/bash
# round 5: extended fuzz + host-buffer soaks of the final build (new seeds) -> appended to profiles/r05_soak.txt
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/soak_r05x; mkdir -p $O
cd $R
{
echo "extended soak, library sha256[:16] $(sha256sum transform360_amd/lib/libTransform360.so | cut -c1-16)"
for m in plane batch plane4 tiny; do
  case $m in plane) n=20000;; batch) n=3000;; plane4) n=1500;; tiny) n=10000;; esac
  echo "mode $m, $n seeds from 100000:"
  timeout 1500 python tests/soak/fuzz_soak.py 100000 $n $m > $O/long_$m.log 2>&1; echo "exit $?"; grep -a "seeds\|mismatch\|differ" $O/long_$m.log | tail -2
done
echo "host-pointer soak (buffers that come and go), 1500 iterations:"
timeout 600 python tests/soak/host_soak.py 1500 2>&1 | tail -1
} 2>&1 | tee $O/soak.txt
