#!/bin/bash
# Round 4, final evidence in one GPU visit (tree must be the committed one: the PMC files carry its source hash).
#   tools/r4_final.sh <outdir> <tag>
OUT=${1:-gpurun_out/r4zz}; TAG=${2:-r04_zz}; mkdir -p $OUT
echo "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo rc=$? >> $OUT/${TAG}_pytest_gpu.txt; tail -4 $OUT/${TAG}_pytest_gpu.txt | cut -c1-200
grep -q "rc=0" $OUT/${TAG}_pytest_gpu.txt || exit 1
echo "== smoke"; timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; tail -1 $OUT/${TAG}_smoke.txt
echo "== evidence"; tools/evidence_run.sh $OUT $TAG 2>&1 | grep -v "^{" | tail -12
echo "== same-box A/B against the round's earlier builds"
L0=FF_LIB_PATH=$PWD/build_var/r3/lib_zz_final.so; L1=FF_LIB_PATH=$PWD/build_var/r4/libff_e62f4da_before_prepass.so; L2=FF_LIB_PATH=$PWD/build_var/r4/libff_b_prepass_template.so
for spec in "c2 1024" "c2 256" "c4 128" "c3 512" "c2 16" "c2 1"; do set -- $spec
  timeout -k 5 600 python tools/pool_sweep.py --isolate --scene $1 --spp $2 --reps 3 "$L0" "$L1" "$L2" "FF_DUMMY=1" 2>&1 | grep -v "^  rep" | sed "s#$PWD/##" | cut -c1-200 | tee -a $OUT/${TAG}_ab_builds.txt
done
echo "== occupancy probe"; timeout -k 5 200 python tools/occupancy_probe.py 64 c2 > $OUT/${TAG}_occupancy_c2.txt 2>&1; tail -12 $OUT/${TAG}_occupancy_c2.txt
echo "== strips"; timeout -k 5 300 python tools/strip_scaling.py 1024 > $OUT/${TAG}_strip_scaling.txt 2>&1; cat $OUT/${TAG}_strip_scaling.txt
echo "== viewer frames"; timeout -k 5 200 python tools/viewer_frame_bench.py > $OUT/${TAG}_viewer_frames.txt 2>&1; cat $OUT/${TAG}_viewer_frames.txt | cut -c1-200
echo "== rank rehearsals (gloo, ranks share the card)"
for n in 2 4 5; do tools/rank_rehearsal.sh $n $OUT/${TAG}_rehearsal_n$n.json 2>&1 | tail -c 600; echo; done
echo "== fuzz"; timeout -k 10 600 python tools/fuzz_parity.py 1000 > $OUT/${TAG}_fuzz_1000.txt 2>&1; tail -2 $OUT/${TAG}_fuzz_1000.txt
echo "== done"
