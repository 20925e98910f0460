#!/bin/bash
# tools/pmc_lds.sh "ENV=.." ... : one SQ pass (LDS activity / bank conflicts, VALU) per setting
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/pl && mkdir -p /tmp/pl
  env $cfg timeout 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES \
      --output-format csv -d /tmp/pl/sq -o sq -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-verify ${BENCH_EXTRA:-} > /tmp/pl/log 2>&1
  echo "== $cfg"
  python "$R/tools/pmc_summary.py" /tmp/pl 2>/dev/null | awk "/kernel=.*(${KFILTER:-remap_tiled})/,/^kernel=zzz/" | grep -E "kernel=|SQ_" | head -${KLINES:-12}
done
