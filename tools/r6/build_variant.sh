#!/bin/bash
# A diagnostic variant of the library: tools/ablate_libs/<name>.so = sot_hip.hip compiled as ONE object with the given SOT_PART mask
# (missing families stubbed) + extra flags, linked with the product's stft / osc / mss objects.  Run with SOT_LIB_PATH=tools/ablate_libs/<name>.so.
#   usage: tools/r6/build_variant.sh <name> <part mask> [flags...]        e.g.  tools/r6/build_variant.sh nowsort 18 -DSOT_WAVE_SORT=0
set -eu
cd "$(dirname "$0")/../.."
PKG=1d-spectral-optimal-transport_amd
name=$1; mask=$2; shift 2
mkdir -p tools/ablate_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DSOT_PART=$mask -DSOT_STUB_MISSING_PARTS "$@" \
    -c -o tools/ablate_libs/$name.o $PKG/csrc/sot_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate_libs/$name.so tools/ablate_libs/$name.o $PKG/csrc/obj/sot_stft.o $PKG/csrc/obj/sot_osc.o $PKG/csrc/obj/sot_mss.o
rm -f tools/ablate_libs/$name.o
echo "built tools/ablate_libs/$name.so"
