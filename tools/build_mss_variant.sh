#!/bin/bash
# tools/ablate_libs/<name>.so = the product library with csrc/sot_mss.hip recompiled with extra flags:
#   tools/build_mss_variant.sh stamps -DMSS_STAMPS
set -e
cd "$(dirname "$0")/.."
name=$1; shift
pkg=1d-spectral-optimal-transport_amd
mkdir -p tools/ablate_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c -o tools/ablate_libs/$name.mss.o $pkg/csrc/sot_mss.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate_libs/$name.so $pkg/csrc/obj/sot_part*.o $pkg/csrc/obj/sot_stft.o $pkg/csrc/obj/sot_osc.o tools/ablate_libs/$name.mss.o
echo tools/ablate_libs/$name.so
