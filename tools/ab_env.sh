#!/bin/bash
# A/B of environment knobs on ONE box with the shipped library: tools/ab_env.sh "bench args" "ENV1=.. ENV2=.." "" ...  (each setting twice, interleaved)
ARGS=$1; shift
for rep in 1 2; do for e in "$@"; do
  echo "[${e:-default}] [$ARGS] $(env $e timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-companion $ARGS | grep -o '"value": [0-9.]*')"
done; done
