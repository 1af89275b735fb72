#!/bin/bash
OUT=gpurun_out/${1:-r4u}; mkdir -p $OUT
for cfg in "FF_TAIL_BLOCKS=1" "FF_DUMMY=0" "FF_TAIL_BLOCKS=3" "FF_TAIL_BLOCKS=2 FF_TAIL_GROUP=32" "FF_TAIL_BLOCKS=1" "FF_DUMMY=0" "FF_TAIL_BLOCKS=3"; do echo "# $cfg" | tee -a $OUT/strips2.txt; env $cfg timeout -k 5 300 python tools/strip_scaling.py 1024 2>&1 | tee -a $OUT/strips2.txt; done
