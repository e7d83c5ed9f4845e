#!/bin/bash
# tools/ablate_libs/<name>.so = the product library with csrc/sot_stft.hip recompiled with extra flags:
#   tools/build_stft_variant.sh nostore -DSOT_STFT_ABLATE_STORE=1
set -e
cd "$(dirname "$0")/.."
name=$1; shift
pkg=1d-spectral-optimal-transport_amd
mkdir -p tools/ablate_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c -o tools/ablate_libs/$name.stft.o $pkg/csrc/sot_stft.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate_libs/$name.so $pkg/csrc/obj/sot_part*.o tools/ablate_libs/$name.stft.o $pkg/csrc/obj/sot_osc.o $pkg/csrc/obj/sot_mss.o
echo tools/ablate_libs/$name.so
