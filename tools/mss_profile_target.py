import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sot_amd import spectra
from sot_amd.losses import MSSLoss
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
a = spectra.harmonic_batch(256, generator=g, device=dev); e0 = spectra.harmonic_batch(256, generator=g, device=dev)
m = MSSLoss(mag_weight=1.0)
for _ in range(30):
    e = e0.detach().requires_grad_(True); m(a, e).backward()
torch.cuda.synchronize()
