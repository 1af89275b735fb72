#!/bin/bash
OUT=gpurun_out/${1:-r4x}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "tail or strip" > $OUT/pytest_tail.log 2>&1; echo rc=$? >> $OUT/pytest_tail.log; tail -4 $OUT/pytest_tail.log | cut -c1-250
grep -q "rc=0" $OUT/pytest_tail.log || exit 1
for spp in 192 256 384 512; do
  timeout -k 5 600 python tools/pool_sweep.py --check --scene c2 --spp $spp --reps 3 "FF_TAIL_BLOCKS=1" "FF_TAIL_BLOCKS=2" "FF_TAIL_BLOCKS=3" 2>&1 | grep -v "^  rep\|check \[" | cut -c1-200 | tee -a $OUT/ab_tail_single.log
done
