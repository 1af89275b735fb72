#!/bin/bash
# Same-box A/B of whole libraries (FF_LIB_PATH), one process per run, interleaved: tools/r4_ab.sh <tag> <reps> lib1 lib2 ...   ("cur" = the tree's library)
TAG=$1; REPS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
CFG=()
for l in "$@"; do if [ "$l" = cur ]; then CFG+=("FF_DUMMY=1"); else CFG+=("FF_LIB_PATH=$PWD/$l"); fi; done
for spec in "c2 256" "c2 1024" "c4 128" "c3 512" "c2 1" "c2 16"; do
  set -- $spec
  timeout -k 5 900 python tools/pool_sweep.py --isolate --scene $1 --spp $2 --reps $REPS "${CFG[@]}" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $OUT/ab.log
done
