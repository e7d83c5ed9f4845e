"""STFT kernels with and without the stored complex spectrum (round 3): forward pair (+ spectrum of the second signal), backward
(recomputing vs from the spectrum), for config 5 (256 clips, n_fft 2048) and the SOT-512 shape (1024 clips, n_fft 512, hop 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sot_amd import _native as nat, spectra
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
def ev(fn, n=80):
    for i in range(20): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n_fft, hop, clips in ((2048, 256, 256), (512, 256, 1024), (512, 128, 256)):
    g = torch.Generator(device=dev).manual_seed(0)
    a = [torch.rand(clips, 4096, device=dev, generator=g) - 0.5 for _ in range(3)]
    b = [torch.rand(clips, 4096, device=dev, generator=g) - 0.5 for _ in range(3)]
    win = spectra._cached_window("flattop", n_fft, dev)
    frames = -(-4096 // hop)
    gm = torch.rand(clips, frames, n_fft // 2 + 1, device=dev, generator=g)
    specs = [nat.stft_mag_forward(b[i], win, n_fft, hop, want_spec=True)[1] for i in range(3)]
    print(f"n_fft {n_fft} hop {hop} clips {clips}: pair fwd {ev(lambda i: nat.stft_mag_forward_pair(a[i % 3], b[i % 3], win, n_fft, hop)):.1f} us, "
          f"pair fwd + spectrum {ev(lambda i: nat.stft_mag_forward_pair(a[i % 3], b[i % 3], win, n_fft, hop, want_spec_b=True)):.1f} us | "
          f"bwd recomputing {ev(lambda i: nat.stft_mag_backward(b[i % 3], win, n_fft, hop, gm)):.1f} us, "
          f"bwd from spectrum {ev(lambda i: nat.stft_mag_backward(b[i % 3], win, n_fft, hop, gm, spec=specs[i % 3])):.1f} us")
