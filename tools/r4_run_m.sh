#!/bin/bash
OUT=gpurun_out/${1:-r4m2}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-250
grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
L1=FF_LIB_PATH=$PWD/build_var/r4/libff_e62f4da_before_prepass.so; L2=FF_LIB_PATH=$PWD/build_var/r4/libff_b_prepass_template.so
for spec in "c2 16 default" "c2 64 default" "c2 256 default" "c2 1024 default" "c3 64 inside" "c3 512 inside" "c2 1024 inside" "c4 128 inside"; do set -- $spec
  timeout -k 5 600 python tools/pool_sweep.py --isolate --scene $1 --spp $2 --camera $3 --reps 3 "$L1" "$L2" "FF_DUMMY=1" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $OUT/ab.log
done
