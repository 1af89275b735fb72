#!/bin/bash
# A library variant of the tree for same-box A/B runs: tools/build_variant.sh <out .so (relative to the repo)> [extra compiler flags]
# (its own object directory; the tree's libfirefly_hip.so is not touched)
cd "$(dirname "$0")/.." || exit 1
OUT=$PWD/$1; shift
mkdir -p "$(dirname "$OUT")"
D=build_x_$(basename "$OUT" .so)
make -s -C gpupathtracer_amd/csrc -j8 OBJDIR=$D OUT="$OUT" EXTRA="$*" 2>&1 | grep -i "error\|warning"
rm -rf gpupathtracer_amd/csrc/$D
ls -la "$OUT"
