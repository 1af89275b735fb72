#!/bin/bash
# Sweep the time-slice / leaf-phase thresholds of the BVH kernel on the benchmark workload at reduced spp.
# Usage (GPU box, repo root): tools/threshold_sweep.sh "8 10 12 14 18" "16 24 32"
for st in $1; do for lt in $2; do
  v=$(FF_SETUP_THRESHOLD=$st FF_LEAF_THRESHOLD=$lt timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp 128 | grep -o '"value": [0-9.]*')
  echo "setup_threshold=$st leaf_threshold=$lt $v"
done; done
