#!/bin/bash
# The reference's own frame rate, by the counters: the 1-spp path-traced 1080p frame of C2 from a camera at rest (stored primary hits kept).
OUT=gpurun_out/${1:-r4y}/pmc_1spp; mkdir -p $OUT
python bench.py --spp 1 --steps 200 --warmup 5 --keep-primary-hits --no-cpu-baseline --no-companion > $OUT/../r04_zz_bench_1spp_at_rest.json 2> $OUT/bench.err; tail -c 600 $OUT/../r04_zz_bench_1spp_at_rest.json
tools/pmc_passes.sh $OUT "sq1 sq2 sq3" --spp 1 --steps 200 --warmup 5 --keep-primary-hits --no-companion
python3 tools/pmc_summary.py $OUT > $OUT/../r04_zz_1spp_pmc_summary.txt 2>&1; head -40 $OUT/../r04_zz_1spp_pmc_summary.txt | cut -c1-160
