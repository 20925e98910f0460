#!/bin/bash
# tools/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ... : run bench.py once per environment setting
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_EXTRA:-} 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step', d['ms_per_step'], 'launch_ms', r['avg_launch_ms'], 'kernel_frac', r['frac'], 'job_frac', d['frac_of_hbm_roofline_whole_job'], 'Mpix/s', d['value'], d['output_checksums'])"
done
