#!/bin/bash
OUT=gpurun_out/${1:-r4knobs}; mkdir -p $OUT
mapfile -t C < <(grep -v '^#' tools/r4_sweep_configs.txt)
for spec in "c2 512" "c4 128"; do set -- $spec
  timeout -k 5 900 python tools/pool_sweep.py --scene $1 --spp $2 --reps 3 --check "${C[@]}" 2>&1 | grep -v "^  rep\|check \[" | cut -c1-200 | tee -a $OUT/knobs.log
done
