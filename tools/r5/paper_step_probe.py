"""The paper's loss step (trainer.py:183-245 with the SOT-2048 YAML) back to back, for rocprofv3 --kernel-trace --stats:
python3 tools/r5/paper_step_probe.py [clips=64] [steps=40] [what=full|modules|mss|sot]   (full: the one-node step of round 6; modules: the same step composed
module by module, fused=False)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from sot_amd import spectra
from sot_amd.losses import MixOfLosses, MSSLoss, Wasserstein1D

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
what = sys.argv[3] if len(sys.argv) > 3 else "full"
dev = torch.device("cuda:0")
mss = MSSLoss(fft_sizes=(2048, 1024, 512, 256, 128, 64), loss_type="L1", mag_weight=1, logmag_weight=0).to(dev)
sot = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True, require_sort=True).to(dev)
mix = MixOfLosses([mss, sot], [0.05, 1]).to(dev)
gen = torch.Generator(device=dev).manual_seed(1000 + clips)
x = spectra.harmonic_batch(clips, generator=gen, device=dev)
hats = [spectra.harmonic_batch(clips, generator=gen, device=dev).requires_grad_(True) for _ in range(2)]
freqs = torch.fft.rfftfreq(2048, d=1.0 / 16000.0).to(dev)
seed = torch.ones((), device=dev)
for i in range(steps + 5):
    e = hats[i % 2]
    e.grad = None
    if what == "mss":
        loss = mss(x, e)
    elif what == "sot":
        loss = spectra.training_step_slice(sot, x, e)
    else:
        loss = spectra.trainer_loss_step(mix, x, e, positions=freqs, fused=(what != "modules"))
    loss.backward(seed)
torch.cuda.synchronize()
print(f"{what} {clips} clips: loss {float(loss):.9g}, |grad| {float(hats[0].grad.abs().sum()):.6g}")
