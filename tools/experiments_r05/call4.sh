#!/bin/bash
# Round 5, call 4: is the 8-frame pipelined step of the native driver (0.0357 ms) faster than bench.py's (0.0399) because of
# Python in the loop, or because a 354 MB input ring still finds lines in the 256 MB Infinity Cache?  Ring sizes 320 .. 1440 MB.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call4; mkdir -p $O
cd $R
run() { echo "== $*" >> $O/native.txt; timeout 120 "$@" 2>&1 | tail -1 >> $O/native.txt; }
for MB in 320 720 1440; do
  for D in 0 2 3 4; do
    P=$([ $D = 0 ] && echo "" || echo "--pipelined $D")
    run examples/t360_multi_gpu --workers 1 --frames 8 --steps 400 --ring-mb $MB $P
  done
done
for D in 0 3; do
  P=$([ $D = 0 ] && echo "" || echo "--pipelined $D")
  run examples/t360_multi_gpu --workers 1 --frames 64 --steps 100 --ring-mb 1440 $P
done
cat $O/native.txt
