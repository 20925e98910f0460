#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
python -m pytest tests -m gpu -x -q -k "lowpass or config3 or fuzz or variants or frame_case" > gpurun_out/gpu_tests.log 2>&1; grep -aE "[0-9]+ (passed|failed)" gpurun_out/gpu_tests.log
export BENCH_EXTRA="--config 3"
tools/sweep.sh "T360_LOWPASS_FRAMES=1" "T360_LOWPASS_FRAMES=2" "T360_LOWPASS_FRAMES=4" "T360_LOWPASS_FRAMES=8" "T360_LOWPASS_FRAMES=16" "T360_LOWPASS_FRAMES=64" "T360_NO_WIDE_LOWPASS=1"
