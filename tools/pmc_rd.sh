#!/bin/bash
# tools/pmc_rd.sh "ENV=.." ... : L2->fabric read requests of the tiled kernel (one PMC pass per setting).
# Exactly the counter set of prof_pmc.sh pass mem1: adding TCC_HIT/MISS to it hung rocprofv3 (4 x 120 s lost).
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/pl && mkdir -p /tmp/pl
  env $cfg timeout 60 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
      --output-format csv -d /tmp/pl/mem1 -o mem1 -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-verify ${BENCH_EXTRA:-} > /tmp/pl/log 2>&1
  echo "== $cfg"
  python "$R/tools/pmc_summary.py" /tmp/pl 2>/dev/null | awk "/kernel=.*remap_tiled/,/^kernel=zzz/" | grep -E "kernel=|TCC_" | head -8
done
