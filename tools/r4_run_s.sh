#!/bin/bash
OUT=gpurun_out/${1:-r4s}; mkdir -p $OUT

cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/diag/one_spp_frames.py > $OUT/trace.log 2>&1; tail -3 $OUT/trace.log
python3 tools/diag/frame_gaps.py $OUT/trace 2 | tee $OUT/frame_gaps.txt | cut -c1-200
head -c 3000 $(ls $OUT/trace/*/*kernel_trace.csv | head -1) > $OUT/trace_head.csv; rm -rf $OUT/trace
