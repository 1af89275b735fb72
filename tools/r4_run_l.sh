#!/bin/bash
OUT=gpurun_out/${1:-r4l}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-250
grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
mapfile -t C < <(grep -v '^#' tools/r4_sweep_configs.txt)
for spec in "c2 256" "c2 1024" "c4 128" "c2 1" "c2 16"; do set -- $spec; timeout -k 5 300 python tools/pool_sweep.py --scene $1 --spp $2 --reps 3 --check "${C[@]}" 2>&1 | grep -v "^  rep" | cut -c1-150 | tee -a $OUT/sweep.log; done
timeout -k 10 500 python tools/fuzz_parity.py 400 > $OUT/fuzz_400.txt 2>&1; tail -4 $OUT/fuzz_400.txt
