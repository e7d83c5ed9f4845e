"""Phase stamps of the fused MSS kernel (variant library built with tools/build_mss_variant.sh stamps -DMSS_STAMPS):
SOT_LIB_PATH=tools/ablate_libs/stamps.so python3 tools/r5/mss_stamps.py [clips] [n_fft ...]
Prints, per workgroup 0..N, the shader clocks between the phase boundaries of its wave 0."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from sot_amd import _native as nat
from sot_amd import spectra

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sizes = tuple(int(v) for v in sys.argv[2:]) or (2048, 1024, 512, 256, 128, 64)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(clips)
x = spectra.harmonic_batch(clips, generator=gen, device=dev)
y = spectra.harmonic_batch(clips, generator=gen, device=dev)
wins = [spectra._cached_window(None, s, dev) for s in sizes]
lib = nat.load()
names = ["tables", "T fetch+win", "T fwd", "T nat+pairs", "V win+fwd", "V nat+pairs+G", "inverse", "OLA+store"]
for rep in range(4):
    nat.mss_loss_and_grad(x, y, sizes, wins, 1.0, 0.0)
    buf = (ctypes.c_ulonglong * (64 * 16))()
    lib.sot_mss_debug_read_stamps.restype = ctypes.c_int
    assert lib.sot_mss_debug_read_stamps(buf, 64 * 16) == 0
print(f"{clips} clips, n_fft {sizes}: shader clocks of wave 0 per phase (last of 4 launches)")
print("wg    " + " ".join(f"{n[:13]:>13s}" for n in names) + "         total")
for wg in list(range(0, 6)) + [20, 40, 63]:
    st = [buf[wg * 16 + i] for i in range(9)]
    d = [st[i + 1] - st[i] for i in range(8)]
    fin = [buf[wg * 16 + i] for i in (11, 12, 13)]
    print(f"{wg:3d}   " + " ".join(f"{v:13d}" for v in d) + f"   {st[8] - st[0]:10d}   finish kernel: start {fin[0] - st[0]:8d} after kernel 1 start, gather {fin[1] - fin[0]:6d}, store {fin[2] - fin[1]:6d}")
