#!/bin/bash
OUT=gpurun_out/${1:-r4t}; mkdir -p $OUT
for spec in "c2 128" "c3 512"; do set -- $spec
  timeout -k 5 600 python tools/pool_sweep.py --check --scene $1 --spp $2 --reps 3 "FF_DUMMY=1" "FF_BVH_OPT_PASSES=0" "FF_BVH_OPT_PASSES=2" "FF_BVH_LEAF=1" "FF_BVH_LEAF=3" "FF_BVH_LEAF=4" "FF_BVH_CTRAV=0.6" "FF_BVH_CTRAV=0.9" "FF_BVH_CTRAV=1.6" "FF_BVH_CTRAV=2.4" "FF_BVH_BINS=16" "FF_BVH_BINS=256" 2>&1 | grep -v "^  rep\|check" | cut -c1-220 | tee -a $OUT/ab_opt2.log
done
