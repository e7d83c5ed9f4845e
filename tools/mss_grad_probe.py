import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sot_amd.losses import MSSLoss
fx = dict(np.load("tests/golden/stft_chain.npz"))
dev = torch.device("cuda:0")
cases = {"paper": dict(mag_weight=1.0, logmag_weight=0.0), "both": dict(mag_weight=1.0, logmag_weight=0.5), "l2": dict(mag_weight=0.7, logmag_weight=0.3, loss_type="L2")}
for tag, kw in cases.items():
    ax = torch.as_tensor(fx["mss_audio_x"]).to(dev)
    ay = torch.as_tensor(fx["mss_audio_y"]).to(dev).requires_grad_(True)
    v = MSSLoss(**kw)(ax, ay); v.backward()
    g = ay.grad.cpu().numpy(); w = fx[f"mss_{tag}_grad_y"]
    ay2 = torch.as_tensor(fx["mss_audio_y"]).to(dev).requires_grad_(True)
    v2 = MSSLoss(**kw)(ax, ay2, dims=[0, 1, 2]); v2.backward()
    g2 = ay2.grad.cpu().numpy()
    rel = lambda a, b: (np.abs(a - b).max() / np.abs(b).max(), np.linalg.norm(a - b) / np.linalg.norm(b), float((a * b).sum() / np.linalg.norm(a) / np.linalg.norm(b)))
    print(tag, "scalar rel", abs(float(v) - float(fx[f"mss_{tag}_loss"])) / float(fx[f"mss_{tag}_loss"]), "hip vs ref (max, l2, cos)", rel(g, w), "torch-gpu-composition vs ref", rel(g2, w), "hip vs torch-gpu", rel(g, g2))
