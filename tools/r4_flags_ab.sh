#!/bin/bash
# Same-box A/B of the kernel object built with different backend options (build_var/r4f/lib_v*.so) against the tree's library.
OUT=gpurun_out/${1:-r4flags}; mkdir -p $OUT
CFG=("FF_DUMMY=1"); for i in 1 2 3 4 5 6; do CFG+=("FF_LIB_PATH=$PWD/build_var/r4f/lib_v$i.so"); done
for spec in "c2 1024" "c2 256" "c4 128"; do set -- $spec
  timeout -k 5 900 python tools/pool_sweep.py --isolate --scene $1 --spp $2 --reps 3 "${CFG[@]}" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $OUT/ab.log
done
