#!/bin/bash
# SQ-side counters of config 4 (Lanczos4) and config 3 (low-pass + gather): who is busy, VALU or LDS?
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
bash tools/prof_pmc.sh $R/gpurun_out/pmc_sq_cfg4 --config 4 > /dev/null 2>&1
bash tools/prof_pmc.sh $R/gpurun_out/pmc_sq_cfg3 --config 3 > /dev/null 2>&1
cat $R/gpurun_out/pmc_sq_cfg4/summary.txt
cat $R/gpurun_out/pmc_sq_cfg3/summary.txt
