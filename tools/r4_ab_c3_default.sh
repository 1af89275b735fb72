#!/bin/bash
# Same box: round 3's final library against the tree on C3 at its own 4 096 spp, the reference's default camera, and C3 at 512 spp.
OUT=${1:-gpurun_out/r4zz}; TAG=${2:-r04_zz}; mkdir -p $OUT
L0=FF_LIB_PATH=$PWD/build_var/r3/lib_zz_final.so
F=$OUT/${TAG}_ab_c3_default_camera.txt
{ echo "# Same box, whole libraries, one process per run, median of 3 (kernel time): round 3's final library against this tree on the two configurations the"
  echo "# headline table quotes from other boxes - C3 at its own 4 096 spp, the reference's default camera - and C3 at 512 spp (a 10 ms tail-mode frame of mostly"
  echo "# dropped items: the noisiest configuration there is)"; } > $F
timeout -k 5 600 python tools/pool_sweep.py --isolate --scene c3 --spp 4096 --reps 3 "$L0" "FF_DUMMY=1" "FF_BVH_OPT_PASSES=0" "FF_NO_PRIMARY_CULL=1" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $F
timeout -k 5 600 python tools/pool_sweep.py --isolate --scene c2 --camera default --spp 1024 --reps 3 "$L0" "FF_DUMMY=1" "FF_BVH_OPT_PASSES=0" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $F
timeout -k 5 600 python tools/pool_sweep.py --isolate --scene c3 --spp 512 --reps 3 "$L0" "FF_DUMMY=1" "FF_BVH_OPT_PASSES=0" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $F
