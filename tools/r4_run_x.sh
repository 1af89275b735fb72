#!/bin/bash
OUT=gpurun_out/${1:-r4x}; mkdir -p $OUT
for spp in 1 2 4; do
echo "# default camera, $spp spp, kept / one-off" | tee -a $OUT/knobs_1spp_d.log
timeout -k 5 600 python tools/pool_sweep.py --keep-primary-hits --camera default --scene c2 --spp $spp --reps 9 "FF_QUEUE_TAIL=8" "FF_QUEUE_TAIL=64" "FF_QUEUE_TAIL=24" 2>&1 | grep -v "^  rep\|check \[" | cut -c1-200 | tee -a $OUT/knobs_1spp_d.log
timeout -k 5 600 python tools/pool_sweep.py --camera default --scene c2 --spp $spp --reps 9 "FF_QUEUE_TAIL=8" "FF_QUEUE_TAIL=64" "FF_QUEUE_TAIL=24" 2>&1 | grep -v "^  rep\|check \[" | cut -c1-200 | tee -a $OUT/knobs_1spp_d.log
done
