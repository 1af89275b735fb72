#!/bin/bash
OUT=gpurun_out/${1:-r4u}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log | cut -c1-250
grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
for cfg in "FF_DUMMY=0" "FF_DUMMY=0" "FF_TAIL_GROUP=16"; do echo "# $cfg" | tee -a $OUT/strips6.txt; env $cfg timeout -k 5 300 python tools/strip_scaling.py 1024 2>&1 | tee -a $OUT/strips6.txt; done
