#!/bin/bash
# One GPU-box visit of round 4: correctness of the job-pool kernel against the lane-owned one, then same-box A/B and knob sweeps.
# usage: tools/r4_session.sh <tag>   (logs under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r4a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== smoke" | tee $OUT/progress.log
timeout -k 5 240 python tools/diag/pool_smoke.py big > $OUT/pool_smoke.log 2>&1 || { echo "smoke FAILED"; tail -30 $OUT/pool_smoke.log; exit 1; }
tail -8 $OUT/pool_smoke.log
echo "== A/B + sweep" | tee -a $OUT/progress.log
mapfile -t CONFIGS < <(grep -v '^#' ${SWEEP_FILE:-tools/r4_sweep_configs.txt})
timeout -k 5 900 python tools/pool_sweep.py --spp 128 --reps 3 --check "${CONFIGS[@]}" \
  > $OUT/sweep_c2.log 2>&1 || { echo "sweep FAILED"; tail -30 $OUT/sweep_c2.log; exit 1; }
grep -v "^  rep" $OUT/sweep_c2.log
echo "== occupancy probes" | tee -a $OUT/progress.log
FF_POOL=1 timeout -k 5 120 python tools/occupancy_probe.py 64 c2 > $OUT/occupancy_c2_pool.txt 2>&1; cat $OUT/occupancy_c2_pool.txt
FF_POOL=0 timeout -k 5 120 python tools/occupancy_probe.py 64 c2 > $OUT/occupancy_c2_lane.txt 2>&1; cat $OUT/occupancy_c2_lane.txt
echo "== other scenes" | tee -a $OUT/progress.log
timeout -k 5 300 python tools/pool_sweep.py --scene c4 --spp 64 --reps 2 --check "FF_POOL=0" "FF_POOL=1" > $OUT/sweep_c4.log 2>&1; grep -v "^  rep" $OUT/sweep_c4.log
timeout -k 5 300 python tools/pool_sweep.py --scene c3 --spp 256 --reps 2 --check "FF_POOL=0" "FF_POOL=1" > $OUT/sweep_c3.log 2>&1; grep -v "^  rep" $OUT/sweep_c3.log
timeout -k 5 200 python tools/pool_sweep.py --scene c2 --spp 1 --reps 3 --check "FF_POOL=0" "FF_POOL=1" > $OUT/sweep_c2_1spp.log 2>&1; grep -v "^  rep" $OUT/sweep_c2_1spp.log
echo "== done" | tee -a $OUT/progress.log
