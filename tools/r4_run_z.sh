#!/bin/bash
OUT=gpurun_out/${1:-r4z}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-300
grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
timeout -k 10 600 python tools/build_bench.py > $OUT/build_bench.txt 2>&1; tail -30 $OUT/build_bench.txt | cut -c1-220
