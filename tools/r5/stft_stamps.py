"""Phase stamps of stft_mag_forward_wavew_kernel (variant library: tools/build_stft_variant.sh stftstamps -DSTFT_STAMPS):
SOT_LIB_PATH=tools/ablate_libs/stftstamps.so python3 tools/r5/stft_stamps.py [clips]   -> shader clocks of wave 0 of the first workgroups per phase"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from sot_amd import _native as nat
from sot_amd import spectra

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
win = spectra._cached_window("flattop", 2048, dev)
gen = torch.Generator(device=dev).manual_seed(clips)
x = spectra.harmonic_batch(clips, generator=gen, device=dev)
lib = nat.load()
for rep in range(4):
    mag, spec = nat.stft_mag_forward(x, win, 2048, 256, want_spec=True)
    nat.stft_mag_backward(x, win, 2048, 256, torch.ones_like(mag), spec=spec)
    buf = (ctypes.c_ulonglong * (64 * 16))()
    lib.sot_stft_debug_read_stamps.restype = ctypes.c_int
    assert lib.sot_stft_debug_read_stamps(buf, 64 * 16) == 0
names = ["tables+barrier", "loads+window", "transform", "bins+stores"]
print("wg   " + " ".join(f"{n:>15s}" for n in names) + "      total")
for wg in (0, 1, 2, 3, 8, 16, 32, 63):
    st = [buf[wg * 16 + i] for i in range(5)]
    if st[0]:
        print(f"{wg:3d}  " + " ".join(f"{st[i + 1] - st[i]:15d}" for i in range(4)) + f"   {st[4] - st[0]:8d}")
names = ["tables+barrier", "pack gradient", "inverse", "window + barrier", "overlap-add", "barrier"]
print("backward (stft_mag_backward_spec_clipw_kernel)")
print("wg   " + " ".join(f"{n:>16s}" for n in names) + "      total")
for wg in (0, 1, 2, 3, 8, 16, 32, 63):
    st = [buf[wg * 16 + 8 + i] for i in range(7)]
    if st[0]:
        print(f"{wg:3d}  " + " ".join(f"{st[i + 1] - st[i]:16d}" for i in range(6)) + f"   {st[6] - st[0]:8d}")
