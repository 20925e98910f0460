#!/bin/bash
# input ring (distinct input > Infinity Cache) vs the same input every step: short steps and small frames
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
for args in "--config 2 --frames 8" "--config 2 --frames 16" "--config 1" "--config 2"; do
 a=$(T360_BENCH_NO_ROTATE=1 python bench.py $args --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('strong_cfg5',{}).get('ms_per_step'), 'verified', d['verified']['max_abs_diff'])")
 b=$(python bench.py $args --no-cpu-baseline --no-host-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('strong_cfg5',{}).get('ms_per_step'), 'verified', d['verified']['max_abs_diff'], d['input_ring']['groups_of_F_frames'])")
 echo "$args: same input every step $a | ring $b"
done
