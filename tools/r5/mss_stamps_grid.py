"""Phase stamps of the fused MSS kernel for workgroups spread over the whole grid (one per scale region): variant library built with
tools/build_mss_variant.sh stamps12 -DMSS_STAMPS -DMSS_STAMP_EVERY=12;  SOT_LIB_PATH=tools/ablate_libs/stamps12.so python3 tools/r5/mss_stamps_grid.py [clips] [every]
(the clock counters of different CUs are not comparable: durations only)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from sot_amd import _native as nat
from sot_amd import spectra

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
every = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sizes = (2048, 1024, 512, 256, 128, 64)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(clips)
x = spectra.harmonic_batch(clips, generator=gen, device=dev)
y = spectra.harmonic_batch(clips, generator=gen, device=dev)
wins = [spectra._cached_window(None, s, dev) for s in sizes]
lib = nat.load()
for rep in range(4):
    nat.mss_loss_and_grad(x, y, sizes, wins, 1.0, 0.0)
    buf = (ctypes.c_ulonglong * (64 * 16))()
    lib.sot_mss_debug_read_stamps.restype = ctypes.c_int
    assert lib.sot_mss_debug_read_stamps(buf, 64 * 16) == 0
names = ["tables", "T fetch+win", "T fwd", "T nat+pairs", "V win+fwd", "V nat+pairs+G", "inverse", "OLA+store"]
print("wg    " + " ".join(f"{n[:13]:>13s}" for n in names) + "         total")
for w in range(64):
    st = [buf[w * 16 + i] for i in range(9)]
    if st[0] == 0 or w % 3:
        continue
    d = [st[i + 1] - st[i] for i in range(8)]
    print(f"{w * every:4d}  " + " ".join(f"{v:13d}" for v in d) + f"   {st[8] - st[0]:10d}")
