#!/bin/bash
# Rehearsal of the driver's multi-GPU launch on a ONE-GPU box: bench.py under torch.distributed.run with N ranks that share the card,
# the gather over gloo (FF_DIST_BACKEND=gloo).  What it exercises is the control flow of the first real node run - rendezvous,
# strip partition, every rank's render, the gather + de-interleave on rank 0, max-over-ranks timing, one JSON line - not xGMI.
# The pool admits at most 6 processes on the card at once, so N <= 5 here (the launcher is the sixth) (the driver's N = 8 runs on a whole node).
#   usage: tools/rank_rehearsal.sh <N> <out.json> [bench args]
N=${1:-2}; OUT=${2:-gpurun_out/bench_n$N.json}; shift 2
PORT=$((20000 + RANDOM % 20000))
FF_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
  bench.py --gpus $N --steps 2 --warmup 1 --spp 64 "$@" > $OUT 2> ${OUT%.json}.err
rc=$?
echo "rank_rehearsal N=$N rc=$rc"; tail -c 1500 $OUT
exit $rc
