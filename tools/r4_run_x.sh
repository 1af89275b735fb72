#!/bin/bash
OUT=gpurun_out/${1:-r4x}; mkdir -p $OUT
for spp in 256 512; do
  timeout -k 5 600 python tools/pool_sweep.py --check --scene c2 --spp $spp --reps 3 "FF_DUMMY=1" "FF_TAIL_GROUP=8" "FF_TAIL_GROUP=4" "FF_TAIL_GROUP=16" 2>&1 | grep -v "^  rep\|check \[" | cut -c1-200 | tee -a $OUT/ab_tail_single2.log
done
