#!/bin/bash
# GPU visit: the whole GPU suite, then same-box A/B of the tree's switches on C2 / C4 / C3 and a short frame.   usage: tools/r4_run_h.sh <tag>
TAG=${1:-r4h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log | cut -c1-250
grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
mapfile -t C < <(grep -v '^#' tools/r4_sweep_configs.txt)
for sc in c2 c4 c3; do timeout -k 5 300 python tools/pool_sweep.py --scene $sc --spp 128 --reps 2 --check "${C[@]}" 2>&1 | grep -v "^  rep" | cut -c1-150 | tee $OUT/sweep_$sc.log; done
timeout -k 5 100 python tools/pool_sweep.py --scene c2 --spp 4 --reps 3 --check "${C[@]}" 2>&1 | grep -v "^  rep" | cut -c1-150 | tee $OUT/sweep_c2_4spp.log
timeout -k 5 100 python tools/pool_sweep.py --scene c2 --spp 1 --reps 3 --check "${C[@]}" "FF_QUEUE_TAIL=2" "FF_QUEUE_TAIL=4" "FF_QUEUE_TAIL=32" 2>&1 | grep -v "^  rep" | cut -c1-150 | tee $OUT/sweep_c2_1spp.log
timeout -k 5 100 python tools/pool_sweep.py --scene c2 --spp 1024 --reps 2 "FF_POOL=0" "FF_QUEUE_TAIL=0" "FF_NO_PRIMARY_REUSE=1" 2>&1 | grep -v "^  rep" | cut -c1-150 | tee $OUT/sweep_c2_1024spp.log
timeout -k 5 100 python tools/diag/last_bounce_census.py 64 2>&1 | tee $OUT/last_bounce_census.txt
