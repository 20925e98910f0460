#!/bin/bash
# Round 5, call 5: 64-frame steps, one stream against pipelined depth 2 / 3, with the SAME 708 MB of input every step
# (what bench.py's headline does) against rings of 2 and 3 batches: is the pipelined gain real for a stream, or does it
# come from overlapping launches finding each other's input lines in the Infinity Cache?  Same box, interleaved.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call5; mkdir -p $O
cd $R
run() { echo "== $*" >> $O/native.txt; timeout 120 "$@" 2>&1 | tail -1 >> $O/native.txt; }
for REP in 1 2; do
for MB in 0 1440 2160; do
  for D in 0 2 3; do
    P=$([ $D = 0 ] && echo "" || echo "--pipelined $D")
    run examples/t360_multi_gpu --workers 1 --frames 64 --steps 100 --ring-mb $MB $P
  done
done
done
cat $O/native.txt
