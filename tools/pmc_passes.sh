#!/bin/bash
# rocprofv3 PMC counters for a bench.py workload, one counter group per run (the MI355X guide: counters in their own passes,
# --kernel-trace only).  Usage (on the GPU box, from the repo root):
#     tools/pmc_passes.sh OUTDIR "sq1 sq2 tcc1 ..." [bench.py args...]
# then: python3 tools/pmc_summary.py OUTDIR > summary.txt
set -u
OUT=$1; shift
PASSES=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A GROUP
GROUP[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
GROUP[sq2]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
GROUP[sq3]="SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INST_LEVEL_VMEM"
GROUP[sqc]="SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES"
GROUP[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum"
GROUP[tcp2]="TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum"
GROUP[tcc1]="FETCH_SIZE"
GROUP[tcc2]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
GROUP[grbm]="GRBM_GUI_ACTIVE"
for name in $PASSES; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc ${GROUP[$name]} --output-format csv -d "$OUT/$name" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
done
