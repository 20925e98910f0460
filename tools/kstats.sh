#!/bin/bash
# tools/kstats.sh "ENV=.." ... : rocprofv3 kernel-trace stats (per-kernel average duration) of a short bench.py run per setting
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/kst && mkdir -p /tmp/kst
  env $cfg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- \
      python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_EXTRA:-} > /tmp/kst/log 2>&1
  echo "== $cfg"
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/kst/**/k_kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if 'remap' in n or 'lowpass' in n: print('   %-60s calls %4s avg %9.1f us'%(n[:60],r['Calls'],float(r['AverageNs'])/1e3))
PY
  tail -1 /tmp/kst/log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   bench launch_ms', d['roofline']['avg_launch_ms'], 'ms/step', d['ms_per_step'])"
done
