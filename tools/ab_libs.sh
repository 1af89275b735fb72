#!/bin/bash
# A/B on ONE box without touching the shipped library: tools/ab_libs.sh "bench args" lib1.so lib2.so ...  (each library twice, interleaved;
# FF_LIB_PATH points the Python binding at the build under test).  Cross-box variation of bench.py is about +-1 %, same-box about 0.1 %.
ARGS=$1; shift
for rep in 1 2; do for so in "$@"; do
  echo "$(basename $so) [$ARGS] $(FF_LIB_PATH=$so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-companion $ARGS | grep -o '"value": [0-9.]*')"
done; done
