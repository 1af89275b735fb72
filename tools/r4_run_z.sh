#!/bin/bash
OUT=gpurun_out/${1:-r4z}; mkdir -p $OUT
for tp in 0 1 2; do echo "# PLOC, FF_TREELET_PASSES=$tp" | tee -a $OUT/reinsert_vs_treelet_ploc.txt; FF_TREELET_PASSES=$tp timeout -k 10 300 python tools/reinsert_ab.py 32 c2,c4 0,4,8 2>&1 | grep "ploc" | cut -c1-200 | tee -a $OUT/reinsert_vs_treelet_ploc.txt; done
