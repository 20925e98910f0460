#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
for f in 1 2 3 8 12 16 20 64; do
  v=$(python bench.py --config 2 --frames $f --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['strong_cfg5']['ms_per_step'], 'verified', d['verified']['max_abs_diff'])")
  echo "frames $f: $v"
done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/suite_pytest.log 2>&1; grep -aE "passed|failed" gpurun_out/suite_pytest.log
