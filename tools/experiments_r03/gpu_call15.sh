#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/c15; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --gather-outputs > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("init", d["init_ms"], "first step", d["first_step_ms"], d["strong_cfg5"], d["gather_outputs"], d["host_abi"])
PY
python bench.py --frames 8 --no-cpu-baseline --no-host-abi > $O/bench8.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$O/bench8.json").read().strip().splitlines()[-1])
print("8 frames:", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["strong_cfg5"])
PY
T360_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-host-abi --gather-outputs > $O/bench_g2.json 2> $O/bench_g2.err; tail -c 700 $O/bench_g2.json
