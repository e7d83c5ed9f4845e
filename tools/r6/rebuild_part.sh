#!/bin/bash
# Recompile the given SOT_PART objects (and/or stft / osc / mss) of the product library and relink it from csrc/obj -- minutes saved
# while iterating on one kernel family.  NOT the product build: finish with `python __graft_entry__.py` (the digest is removed here so
# that a partial rebuild is never mistaken for the product library).   usage: tools/r6/rebuild_part.sh 16 [2 ...] [-- extra hipcc flags]
set -eu
cd "$(dirname "$0")/../.."
PKG=1d-spectral-optimal-transport_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
parts=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; parts+=("$1"); shift; done
rm -f $PKG/libsot_hip.so.digest
pids=()
for p in "${parts[@]}"; do
  case "$p" in
    stft|osc|mss) /opt/rocm/bin/hipcc $FLAGS "${extra[@]}" -c -o $PKG/csrc/obj/sot_$p.o $PKG/csrc/sot_$p.hip & ;;
    *) /opt/rocm/bin/hipcc $FLAGS "${extra[@]}" -DSOT_PART=$p -c -o $PKG/csrc/obj/sot_part$p.o $PKG/csrc/sot_hip.hip & ;;
  esac
  pids+=($!)
done
for pid in "${pids[@]}"; do wait $pid; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libsot_hip.so.tmp $PKG/csrc/obj/sot_part*.o $PKG/csrc/obj/sot_stft.o $PKG/csrc/obj/sot_osc.o $PKG/csrc/obj/sot_mss.o
mv $PKG/libsot_hip.so.tmp $PKG/libsot_hip.so
if [ ${#extra[@]} -eq 0 ]; then python3 -c "import sys; sys.path.insert(0, '.'); import sot_amd; open(sot_amd.build.DIGEST, 'w').write(sot_amd.build.source_digest() + chr(10))"; fi   # iteration aid: the caller vouches that every part a changed header touches was recompiled
echo "relinked $PKG/libsot_hip.so (no digest: run python __graft_entry__.py before the round's final runs)"
