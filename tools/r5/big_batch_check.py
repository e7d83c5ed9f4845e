"""The two-launch MSSLoss on a batch beyond the finish kernel's grid cap (4200 clips x 4096 samples: 33 600 (clip, range) blocks on 16 384
workgroups, 1 GB of scratch): per-clip losses and gradients bit-identical to the batch evaluated in chunks of 1500 clips, the scalar form consistent
with them.  python3 tools/r5/big_batch_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sot_amd.losses as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
B = 4200
x = torch.randn(B, 4096, device=dev, generator=g); y = x + 0.3 * torch.randn(B, 4096, device=dev, generator=g)
mod = L.MSSLoss(mag_weight=1.0).to(dev)
def per_clip(lo, hi):
    yd = y[lo:hi].clone().requires_grad_(True)
    v = mod(x[lo:hi], yd, dims=(1, 2)); v.sum().backward()
    return v.detach(), yd.grad
whole = per_clip(0, B)
parts = [per_clip(i, min(i + 1500, B)) for i in range(0, B, 1500)]
print("per-clip losses equal:", torch.equal(whole[0], torch.cat([p[0] for p in parts])), " gradients equal:", torch.equal(whole[1], torch.cat([p[1] for p in parts])))
yd = y.clone().requires_grad_(True); v = mod(x, yd); v.backward()
print("scalar loss", float(v), "mean of per-clip", float(whole[0].double().mean()), "grad finite", bool(torch.isfinite(yd.grad).all()),
      "scalar-grad vs per-clip-grad/B", float((yd.grad - whole[1] / B).abs().max() / (whole[1] / B).abs().max()))
