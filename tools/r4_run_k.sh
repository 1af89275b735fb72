#!/bin/bash
OUT=gpurun_out/r4k; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-250
grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
tools/r4_ab.sh r4k 3 build_var/r4/libff_e62f4da_before_prepass.so build_var/r4/libff_a_prepass_tail.so cur
