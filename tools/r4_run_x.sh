#!/bin/bash
OUT=gpurun_out/${1:-r4x}; mkdir -p $OUT
for spec in "c2 1024 default" "c3 4096 inside" "c3 1024 inside"; do set -- $spec
  timeout -k 5 600 python tools/pool_sweep.py --check --scene $1 --spp $2 --camera $3 --reps 3 "FF_DUMMY=1" "FF_TAIL_GROUP=16" "FF_TAIL_GROUP=16,FF_TAIL_BLOCKS=2" "FF_TAIL_GROUP=8,FF_TAIL_BLOCKS=2" "FF_TAIL_GROUP=8,FF_TAIL_BLOCKS=3" 2>&1 | grep -v "^  rep\|check \[" | cut -c1-200 | tee -a $OUT/ab_tail_culled.log
done
