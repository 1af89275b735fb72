#!/bin/bash
# A/B on ONE box: tools/ab_bench.sh "env-or-empty" lib1.so lib2.so ...   (each library benchmarked twice, interleaved)
# Cross-box variation of bench.py is about +-1 %, same-box repeatability about 0.1 %.
cp gpupathtracer_amd/libfirefly_hip.so /tmp/ab_orig.so
for rep in 1 2; do for so in "$@"; do
  cp "$so" gpupathtracer_amd/libfirefly_hip.so
  echo "$(basename $so) $(timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --spp 256 | grep -o '"value": [0-9.]*')"
done; done
cp /tmp/ab_orig.so gpupathtracer_amd/libfirefly_hip.so
