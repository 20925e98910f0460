#!/bin/bash
# Round 5, call 10: the default line with the native-driver leg; the two-rank gloo rehearsal of bench.py (two ranks on the one
# GPU); the GPU suite.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call10; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -a -o -E "[0-9]+ passed[^\n]{0,60}|[0-9]+ failed[^\n]{0,60}" $O/pytest.log | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 700 $O/bench_default.json; echo
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("native_driver"), indent=1)[:1800])
print(d["strong_cfg5"]["projected_8_gpus"])
PY
T360_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err; tail -c 600 $O/bench_gpus2_gloo.err; head -c 500 $O/bench_gpus2_gloo.json; echo
python - <<PY
import json
d = json.loads(open("$O/bench_gpus2_gloo.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("native_driver"), indent=1)[:1200])
PY
