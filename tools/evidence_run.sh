#!/bin/bash
# The round's evidence in one GPU call: rocprofv3 kernel-trace statistics and PMC counters of the headline workload (C2) and
# of C4 (the config that leaves LDS), then the plain bench line with the roofline object filled from the fresh PMC file.
# Usage (GPU box, repo root): tools/evidence_run.sh OUTDIR TAG
set -u
OUT=$1; TAG=$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {  # name, bench args...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python3 bench.py --no-cpu-baseline --no-companion "$@" > "$OUT/prof_$name.log" 2>&1
  echo "stats $name rc=$?"
  find "$OUT/prof_$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_${name}_kernel_stats.csv" \;
  rm -rf "$OUT/prof_$name"
}
stats c2 --steps 3 --warmup 1
tools/pmc_passes.sh "$OUT/pmc_c2" "sq1 sq2 sq3 sqc tcp tcc1 tcc2 grbm" --steps 1 --warmup 1 --no-companion
python3 tools/pmc_summary.py "$OUT/pmc_c2" > "$OUT/${TAG}_c2_pmc_summary.txt"
python3 tools/pmc_to_json.py "$OUT/pmc_c2" "$OUT/${TAG}_c2_pmc.json" "round 4 kernel (round 3 + primary hits stored per pixel by a pre-pass, exact cull, as-asked tail zone of the work queue)"
rm -rf "$OUT"/pmc_c2/*/
stats c4 --scene c4 --steps 3 --warmup 1
tools/pmc_passes.sh "$OUT/pmc_c4" "sq1 sq2 sq3 tcp tcc1 tcc2 grbm" --scene c4 --steps 1 --warmup 1 --no-companion
python3 tools/pmc_summary.py "$OUT/pmc_c4" > "$OUT/${TAG}_c4_pmc_summary.txt"
python3 tools/pmc_to_json.py "$OUT/pmc_c4" "$OUT/${TAG}_c4_pmc.json" "round 4 kernel, C4: the config whose tree does not fit LDS"
rm -rf "$OUT"/pmc_c4/*/
cp "$OUT/${TAG}_c2_pmc.json" "$OUT/${TAG}_c4_pmc.json" profiles/  # (on the GPU box: so that the bench lines below find them)
timeout -k 10 400 python3 bench.py --steps 5 --warmup 1 > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
tail -c 3000 "$OUT/${TAG}_bench.json"
timeout -k 10 300 python3 bench.py --scene c4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_c4_bench.json" 2>> "$OUT/bench.err"
timeout -k 10 300 python3 bench.py --scene c3 --spp 4096 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_c3_bench.json" 2>> "$OUT/bench.err"
tail -c 700 "$OUT/${TAG}_c4_bench.json"; tail -c 700 "$OUT/${TAG}_c3_bench.json"
