"""Where the log-magnitude term of MSSLoss loses accuracy on clips shorter than the frame (tools/r5/fuzz_mss.py flags them): magnitudes and
gradient of the HIP chain and of torch float32 against float64 for a 64-sample clip under n_fft 1024 / 2048 and for a 4096-sample clip.
Observed: equal absolute errors (2e-7 of the peak); on the 64-sample / n_fft 1024 case the HIP transform's error at the SMALL bins is 6 x
torch's relative to the bin (5e-6 against 8e-7).  ROUND 6: the gradient figure of round 5 (2.4e-5 against 1.9e-6 of the gradient's norm) was
the YARDSTICK's window -- hann computed on the CPU, where the module and the reference (utils.py:200-201) compute it on the audio's device;
the first taps differ by 3e-6 ... 9e-5 relative (tools/r6/mss_debug.py).  Against the device's window: 2.6e-6 against 1.2e-6.
python3 tools/r5/logmag_case.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sot_amd import losses, spectra
import fuzz_mss as fm
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
for samples, size in ((64, 1024), (64, 2048), (4096, 1024)):
    x = torch.randn(1, samples, device=dev, generator=g); y = x + 0.3 * torch.randn(1, samples, device=dev, generator=g)
    hop = size // 4
    mag = spectra.stft_magnitude(y, size, hop, None).double()
    win = torch.hann_window(size, device=dev)
    def mg(a, dt):
        a = spectra.end_padded(a.to(dt), size, hop)
        return torch.stft(a, n_fft=size, hop_length=hop, win_length=size, window=win.to(dt), center=False, normalized=True, return_complex=True).abs().permute(0, 2, 1)
    m64, m32 = mg(y, torch.float64), mg(y, torch.float32).double()
    live = m64 > 1e-5
    print(f"samples {samples} n_fft {size}: frames {mag.shape[1]}, |V| range of live bins {float(m64[live].min()):.3g} .. {float(m64.max()):.3g}")
    print("   HIP   magnitude: max abs err / peak", float((mag - m64).abs().max() / m64.max()), " max REL err over live bins", float(((mag - m64).abs() / m64)[live].max()))
    print("   torch magnitude: max abs err / peak", float((m32 - m64).abs().max() / m64.max()), " max REL err over live bins", float(((m32 - m64).abs() / m64)[live].max()))
    mod = losses.MSSLoss(fft_sizes=(size,), loss_type="L2", mag_weight=0.0, logmag_weight=1.0).to(dev)
    res = {}
    for dt in (torch.float64, torch.float32):
        yy = y.to(dt).clone().requires_grad_(True); v = fm.mss_torch(mod, x, yy, None, dt); v.backward(); res[dt] = yy.grad.double()
    yy = y.clone().requires_grad_(True); mod(x, yy).backward()
    n = torch.linalg.norm(res[torch.float64])
    print("   gradient err (norm): HIP", float(torch.linalg.norm(yy.grad.double() - res[torch.float64]) / n), " torch float32", float(torch.linalg.norm(res[torch.float32] - res[torch.float64]) / n))
