#!/bin/bash
# Round 5, call 9: the 4-wave plan of short steps (8 frames, config 2) with 128x8 strips where their row fragments touch
# fewer lines than 64x16 tiles do (PlanOptions::strip_pct with cost_lines); native driver, instrumented build, from HBM.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call9; mkdir -p $O
cd $R
run() { echo "== $*" >> $O/native.txt; timeout 120 "$@" 2>&1 | tail -1 >> $O/native.txt; }
for REP in 1 2; do
for ENV in "T360_X=0" "T360_COST_LINES=1" "T360_COST_LINES=1 T360_STRIPS=100" "T360_COST_LINES=1 T360_STRIPS=110" "T360_STRIPS=100" "T360_STRIPS=120"; do
  for D in 0 2; do
    P=$([ $D = 0 ] && echo "" || echo "--pipelined $D")
    run env $ENV examples/t360_multi_gpu_instr --workers 1 --frames 8 --steps 400 --ring-mb 1440 $P
  done
done
done
grep -A1 "^==" $O/native.txt | grep -v "^--" | paste - - | sed 's/examples\/t360_multi_gpu_instr --workers 1 --frames 8 --steps 400 --ring-mb 1440//; s/1 worker(s) on 1 device(s), weak scaling, compute only//' | cut -c1-200
