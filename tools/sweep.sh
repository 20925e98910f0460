#!/bin/bash
# tools/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ... : run bench.py once per environment setting of the INSTRUMENTED library
# (make -C transform360_amd/csrc instr); prints ms per step.  Development only.
R=$(cd "$(dirname "$0")/.." && pwd)
export T360_LIB=${T360_LIB:-$R/transform360_amd/lib/libTransform360_instr.so}
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg T360_BENCH_ALLOW_INSTRUMENTED=1 timeout 300 python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-host-abi ${BENCH_EXTRA:-} 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step', d['ms_per_step'], 'launch_ms', r['avg_launch_ms'], 'kernel_frac', r['frac'], 'Mpix/s', d['value'], d['output_checksums'])"
done
