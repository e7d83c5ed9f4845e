"""STFT-magnitude producer: HIP kernels vs the torch.stft path (rocFFT), forward and forward+backward; 256 clips x 4096
samples, n_fft 2048 / hop 256 (config 5) and n_fft 512 / hop 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sot_amd import spectra
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
audio = spectra.harmonic_batch(256, generator=g, device=dev)

def ev(fn, n=50, warm=20):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for n_fft, hop in ((2048, 256), (512, 128)):
    frames = -(-audio.shape[1] // hop); bins = n_fft // 2 + 1
    up = torch.rand(256, frames, bins, device=dev)
    def fwd(fn):
        with torch.no_grad(): fn(audio, n_fft, hop)
    def fb(fn):
        a = audio.detach().requires_grad_(True)
        (fn(a, n_fft, hop) * up).sum().backward()
    th, tt = ev(lambda: fwd(spectra.stft_magnitude)), ev(lambda: fwd(spectra.stft_magnitude_torch))
    bh, bt = ev(lambda: fb(spectra.stft_magnitude)), ev(lambda: fb(spectra.stft_magnitude_torch))
    out_mb = 256 * frames * bins * 4 / 1e6
    print(f"n_fft {n_fft} hop {hop}: {256 * frames} frames, {out_mb:.1f} MB of magnitudes: forward HIP {th:7.1f} us  torch {tt:7.1f} us | "
          f"forward+backward (incl. autograd) HIP {bh:7.1f} us  torch {bt:7.1f} us")

# MSSLoss (6 FFT sizes, magnitude L1: the paper's setting), forward + backward w.r.t. the estimate: HIP module vs the same
# composition on torch.stft
from sot_amd.losses import MSSLoss
est = spectra.harmonic_batch(256, generator=g, device=dev)
mss = MSSLoss(mag_weight=1.0)
def mss_hip():
    e = est.detach().requires_grad_(True)
    mss(audio, e).backward()
def mss_torch():
    e = est.detach().requires_grad_(True)
    loss = 0.0
    for size in mss.fft_sizes:
        t = spectra.stft_magnitude_torch(audio, size, size // 4, None)
        v = spectra.stft_magnitude_torch(e, size, size // 4, None)
        loss = loss + (t - v).abs().mean()
    loss.backward()
print(f"MSSLoss fwd+bwd, 256 clips x 4096 samples, 6 scales: HIP {ev(mss_hip, 20, 10):7.1f} us   torch.stft composition {ev(mss_torch, 20, 10):7.1f} us")

# the same MSSLoss step replayed from one HIP graph (kernel-bound figure: no Python / autograd / launch overhead per step)
e_static = est.detach().requires_grad_(True)
def mss_static():
    e_static.grad = None
    mss(audio, e_static).backward()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): mss_static()
torch.cuda.current_stream().wait_stream(side)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    mss_static()
print(f"MSSLoss fwd+bwd replayed from one HIP graph: {ev(gr.replay, 50, 10):7.1f} us")
