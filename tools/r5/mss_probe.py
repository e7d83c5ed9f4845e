"""Timing of the two-launch MSSLoss (sot_mss_loss_and_grad) by scale and batch: python3 tools/r5/mss_probe.py [reps]
(SOT_LIB_PATH selects a library variant).  Prints us per call: all six scales, each scale alone, with and without the gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from sot_amd import _native as nat
from sot_amd import spectra

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
SIZES = (2048, 1024, 512, 256, 128, 64)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for clips in (64, 256):
    gen = torch.Generator(device=dev).manual_seed(clips)
    x = spectra.harmonic_batch(clips, generator=gen, device=dev)
    y = spectra.harmonic_batch(clips, generator=gen, device=dev)
    for sizes in (SIZES,) + tuple((s,) for s in SIZES):
        wins = [spectra._cached_window(None, s, dev) for s in sizes]
        row = []
        for grad in (True, False):
            row.append(timed(lambda: nat.mss_loss_and_grad(x, y, sizes, wins, 1.0, 0.0, 1e-5, False, False, grad)))
        print(f"{clips:4d} clips, n_fft {str(sizes):40s} loss+grad {row[0]:8.1f} us   loss only {row[1]:8.1f} us", flush=True)
