#!/bin/bash
# Round 5, call 14: the launcher form of a multi-GPU run rehearsed with ONE rank on the one GPU, RCCL for real
# (T360_FORCE_DIST=1: process group, context broadcast, barriers, all_reduce, all_gather, gather / scatter, store wait).
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05_call14; mkdir -p $O
cd $R
T360_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 20 --warmup 5 --gather-outputs --scatter-inputs > $O/line.json 2> $O/err.txt
echo "rc $?"; tail -c 1500 $O/err.txt | grep -v "amdgpu.ids\|socket.cpp" | tail -8
python - <<PY
import json
d = json.loads([l for l in open("$O/line.json").read().strip().splitlines() if l.startswith('{"metric"')][-1])
print(d["value"], d["ms_per_step"], d["n_gpus"], d["verified"]["max_abs_diff"], d["verified"].get("all_ranks_ok"), d["output_checksums"])
print("gather", d["gather_outputs"]["collective"], d["gather_outputs"]["ms_per_step"], "| scatter", d["scatter_gather"]["collective"], d["scatter_gather"]["ms_per_step"])
print("native", json.dumps(d.get("native_driver"))[:300])
PY
