"""rocprofv3 target: the two-launch MSSLoss without / with the gradient, 40 calls each: python3 tools/r5/mss_nograd_probe.py [clips]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sot_amd import _native as nat
from sot_amd import spectra
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(clips)
x = spectra.harmonic_batch(clips, generator=gen, device=dev)
y = spectra.harmonic_batch(clips, generator=gen, device=dev)
sizes = (2048, 1024, 512, 256, 128, 64)
wins = [spectra._cached_window(None, s, dev) for s in sizes]
for grad in (False,):
    for _ in range(40):
        nat.mss_loss_and_grad(x, y, sizes, wins, 1.0, 0.0, 1e-5, False, False, grad)
torch.cuda.synchronize()
