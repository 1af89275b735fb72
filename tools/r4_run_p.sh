#!/bin/bash
OUT=gpurun_out/${1:-r4p}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log | cut -c1-250
grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
L2=FF_LIB_PATH=$PWD/build_var/r4/libff_b_prepass_template.so
for spec in "c2 1" "c2 2" "c2 16" "c2 1024" "c4 128"; do set -- $spec
  timeout -k 5 600 python tools/pool_sweep.py --isolate --scene $1 --spp $2 --reps 3 "$L2" "FF_DUMMY=1" "FF_REUSE_MIN_SPP=1" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $OUT/ab.log
done
echo "# with the stored hits kept between frames (a camera at rest)" | tee -a $OUT/ab.log
for spec in "c2 1" "c2 4"; do set -- $spec
  timeout -k 5 600 python tools/pool_sweep.py --isolate --keep-primary-hits --scene $1 --spp $2 --reps 3 "$L2" "FF_DUMMY=1" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $OUT/ab.log
done
timeout -k 5 200 python tools/viewer_frame_bench.py > $OUT/viewer_frames.txt 2>&1; cat $OUT/viewer_frames.txt | cut -c1-200
