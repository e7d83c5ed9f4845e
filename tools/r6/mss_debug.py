"""Round 6: the 'accuracy hole' of MSSLoss's log-magnitude term on clips shorter than a frame (round-5 review), taken apart: the float64
yardstick of tools/r5/logmag_case.py used the hann window torch computes on the CPU, the module (like the reference, utils.py:200-201:
torch.hann_window(..., device=audio.device)) the one torch computes ON THE GPU.  The two differ in the last bits, and a 64-sample clip only
meets the window's first taps (1e-5 ... 4e-2), where 0.5 - 0.5 cos(2 pi k / N) in float32 carries relative errors of 1e-5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "r5"))
import torch
from sot_amd import _native as nat, spectra
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
for samples, size in ((64, 1024), (64, 2048), (100, 512), (4096, 1024)):
    x = torch.randn(3, samples, device=dev, generator=g); y = x + 0.3 * torch.randn(3, samples, device=dev, generator=g)
    hop = size // 4
    w_dev, w_cpu = torch.hann_window(size, device=dev), torch.hann_window(size).to(dev)
    live = w_cpu[:samples] > 0
    print(f"samples {samples} n_fft {size}: hann on the GPU vs on the CPU, first {samples} taps: max relative difference {float(((w_dev - w_cpu).abs()[:samples][live] / w_cpu[:samples][live]).max()):.2e}")

    def chain(win, dt):
        yy = y.to(dt).clone().requires_grad_(True)
        def mag(a):
            a = spectra.end_padded(a.to(dt), size, hop)
            return torch.stft(a, n_fft=size, hop_length=hop, win_length=size, window=win.to(dt), center=False, normalized=True, return_complex=True).abs()
        eps = torch.tensor(1e-5, dtype=dt, device=dev)
        t, v = mag(x), mag(yy)
        d = torch.log(torch.where(t <= eps, eps, t)) - torch.log(torch.where(v <= eps, eps, v))
        torch.mean(d ** 2).backward()
        return yy.grad.double()
    _, ghip = nat.mss_loss_and_grad(x, y, (size,), [w_dev], 0.0, 1.0, l2=True)
    for name, win in (("CPU-computed window", w_cpu), ("GPU-computed window (what the module and the reference on this device use)", w_dev)):
        g64, g32 = chain(win, torch.float64), chain(win, torch.float32)
        n = torch.linalg.norm(g64)
        print(f"   yardstick with the {name}: HIP two-launch kernel {float(torch.linalg.norm(ghip.double() - g64) / n):.2e}   torch.stft float32 {float(torch.linalg.norm(g32 - g64) / n):.2e}")
