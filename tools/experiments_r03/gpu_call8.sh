#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c8; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "variants or config2 or shipped or batch" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
python bench.py --no-cpu-baseline --no-host-abi > $O/bench64.json 2> $O/bench64.err; tail -c 1500 $O/bench64.json | head -c 1500; echo; tail -3 $O/bench64.err
python bench.py --no-cpu-baseline --no-host-abi --frames 8 > $O/bench8.json 2> $O/bench8.err; python - <<PY
import json
for f in ("$O/bench64.json","$O/bench8.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["kernel"], d["verified"])
    except Exception as e: print(f, "ERR", e)
PY
