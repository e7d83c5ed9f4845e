"""STFT-magnitude forward (n_fft 2048, hop 256, flattop) by batch: one signal, the pair launch, with the stored spectrum; us per call, best of 5 rounds.
python3 tools/r5/stft_fwd_probe.py [clips...]   (SOT_LIB_PATH selects a library variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from sot_amd import _native as nat
from sot_amd import spectra

dev = torch.device("cuda:0")
win = spectra._cached_window("flattop", 2048, dev)


def timed(fn, reps=100):
    best = 1e9
    for _ in range(5):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


for clips in [int(v) for v in sys.argv[1:]] or [16, 32, 64, 96, 128, 180]:
    gen = torch.Generator(device=dev).manual_seed(clips)
    x = spectra.harmonic_batch(clips, generator=gen, device=dev)
    y = spectra.harmonic_batch(clips, generator=gen, device=dev)
    frames = nat.stft_mag_forward(x, win, 2048, 256).shape[1]
    one = timed(lambda: nat.stft_mag_forward(x, win, 2048, 256))
    spec = timed(lambda: nat.stft_mag_forward(x, win, 2048, 256, want_spec=True))
    pair = timed(lambda: nat.stft_mag_forward_pair(x, y, win, 2048, 256, want_spec_b=True))
    print(f"{clips:4d} clips ({clips * frames:5d} frames): one signal {one:6.1f} us, with spectrum {spec:6.1f} us, pair + spectrum ({2 * clips * frames} frames) {pair:6.1f} us", flush=True)
