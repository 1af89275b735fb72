#!/bin/bash
OUT=gpurun_out/${1:-r4x}; mkdir -p $OUT
timeout -k 5 900 python tools/pool_sweep.py --check --scene c4 --spp 128 --reps 2 "FF_DUMMY=1" "FF_BVH_LEAF=3" "FF_BVH_LEAF=4" "FF_BVH_LEAF=6" "FF_BVH_LEAF=8" "FF_BVH_CTRAV=0.8" "FF_BVH_CTRAV=2.0" "FF_BVH_LEAF=4,FF_BVH_CTRAV=2.0" "FF_BVH_BINS=32" 2>&1 | grep -v "^  rep\|check \[" | cut -c1-200 | tee -a $OUT/ab_c4_builder.log
