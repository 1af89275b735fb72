#!/bin/bash
# Same-box A/B of whole libraries: tools/r4_ab_libs.sh <out dir under gpurun_out> <lib> [<lib> ...]   (first = baseline)
# SPECS="c2:128 c2:1024 ..." overrides the workloads; TESTS=1 runs the GPU suite on the tree's own library first.
OUT=gpurun_out/$1; shift; mkdir -p $OUT
if [ -n "$TESTS" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo rc=$? >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log | cut -c1-250
  grep -q "rc=0" $OUT/pytest_gpu.log || exit 1
fi
CFG=(); for l in "$@"; do CFG+=("FF_LIB_PATH=$PWD/$l"); done
for spec in ${SPECS:-c2:128 c2:1024 c2:16 c2:1 c4:128}; do
  timeout -k 5 900 python tools/pool_sweep.py --isolate --scene ${spec%%:*} --spp ${spec##*:} --reps ${REPS:-3} "${CFG[@]}" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $OUT/ab.log
done
