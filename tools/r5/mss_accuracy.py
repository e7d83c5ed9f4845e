"""Accuracy of the two-launch MSSLoss against the float64 composition, per scale, next to the reference's own float32 error:
[SOT_LIB_PATH=variant.so] python3 tools/r5/mss_accuracy.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_mss_fused as T
from sot_amd.losses import MSSLoss

dev = torch.device("cuda:0")
for batch, samples, seed in ((5, 4096, 4096), (4, 4096, 11), (2, 9000, 9000)):
    x, y = T._clips(batch, samples, seed)
    for sizes in (T.SIZES,) + tuple((s,) for s in T.SIZES):
        mod = MSSLoss(fft_sizes=sizes, mag_weight=1.0)
        w64, g64 = T._reference(mod, x, y)
        w32, g32 = T._reference(mod, x, y, dtype=torch.float32)
        yd = y.to(dev).requires_grad_(True)
        got = mod(x.to(dev), yd)
        got.backward()
        g = yd.grad.cpu().double()
        n = torch.linalg.norm(g64)
        print(f"{batch}x{samples} n_fft {str(sizes):38s} loss err hip {abs(float(got) - float(w64)) / float(w64):.2e} ref32 {abs(float(w32) - float(w64)) / float(w64):.2e}   "
              f"grad L2 err hip {float(torch.linalg.norm(g - g64) / n):.2e} ref32 {float(torch.linalg.norm(g32.double() - g64) / n):.2e}   "
              f"max err hip {float((g - g64).abs().max() / g64.abs().max()):.2e} ref32 {float((g32.double() - g64).abs().max() / g64.abs().max()):.2e}", flush=True)
