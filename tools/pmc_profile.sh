#!/bin/bash
# Collect rocprofv3 PMC counters for the trace kernel in separate passes (one counter group per run, as the MI355X
# guide prescribes).  Usage (on the GPU box, from the repo root):  tools/pmc_profile.sh OUTDIR [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  counters=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/$name" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
pass sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "$@"
pass sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" "$@"
pass tcc1 "FETCH_SIZE" "$@"
pass tcc2 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "$@"
pass grbm "GRBM_GUI_ACTIVE" "$@"
