#!/bin/bash
# round 4, GPU call 25: rotation -- every 64-frame run starts at the frame a wall clock is on and wraps around
# (T360_ROTATE = 100 MHz ticks per frame), so that resident workgroups are near the same frame without anyone waiting.
# Model (tests/plan_sim/l2replay.py with T360_SIM_ROTATE on the trace of call 3): line reads 1.509x -> 1.42x.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r04c25; mkdir -p $O
cd $R
export BENCH_EXTRA="--steps 20"
# verified once (bench.py compares frames with the oracle unless --no-verify)
T360_ROTATE=96 T360_LIB=$R/transform360_amd/lib/libTransform360_instr.so T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-host-abi --no-two-streams 2>&1 | tail -1 | cut -c1-700 | tee $O/rot96_verified.json
for rep in 1 2; do
tools/sweep.sh "T360_ROTATE=0" "T360_ROTATE=80" "T360_ROTATE=96" "T360_ROTATE=110" "T360_ROTATE=130" "T360_ROTATE=1000000" 2>&1 | tee -a $O/sweep.txt
done
