#!/bin/bash
# the GPU suite + the default bench line (what the driver runs at round end)
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/suite; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 900 $O/bench.json
