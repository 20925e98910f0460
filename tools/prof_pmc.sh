#!/bin/bash
# tools/prof_pmc.sh -- rocprofv3 PMC passes for bench.py (run on the GPU box via gpurun).
# Counters are collected in separate passes (SQ: 8 slots, TCC: FETCH_SIZE costs 3, WRITE_SIZE 2;
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), each with --kernel-trace only.
# usage: tools/prof_pmc.sh <outdir> [bench args...]
set -u
OUT=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() { # name counters...
  local name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- \
      python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-host-abi --no-two-streams --no-native "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
BENCH_ARGS=("$@")
if [ "${PMC_MEM:-0}" != "only" ]; then
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR
pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
fi
if [ "${PMC_MEM:-0}" = "1" ] || [ "${PMC_MEM:-0}" = "only" ]; then
pass mem1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass mem2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
pass mem3 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum
# NOTE: TA_* counters hang rocprofv3 on this pool (two 600 s timeouts, 2026-09-23): never request them.
if [ "${PMC_MEM:-0}" = "1" ]; then
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH
fi
fi
python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
