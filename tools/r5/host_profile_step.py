"""Host-side profile (cProfile) of the paper's loss step launched eagerly: python3 tools/r5/host_profile_step.py [clips]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sot_amd import spectra
from sot_amd.losses import MixOfLosses, MSSLoss, Wasserstein1D
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
mss = MSSLoss(fft_sizes=(2048, 1024, 512, 256, 128, 64), loss_type="L1", mag_weight=1, logmag_weight=0).to(dev)
sot = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True, require_sort=True).to(dev)
mix = MixOfLosses([mss, sot], [0.05, 1]).to(dev)
gen = torch.Generator(device=dev).manual_seed(1000 + clips)
x = spectra.harmonic_batch(clips, generator=gen, device=dev)
hats = [spectra.harmonic_batch(clips, generator=gen, device=dev).requires_grad_(True) for _ in range(2)]
seed = torch.ones((), device=dev)

def step(i):
    e = hats[i % 2]
    e.grad = None
    spectra.trainer_loss_step(mix, x, e).backward(seed)

for i in range(50):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300):
    step(i)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{clips} clips: {1e6 * host / 300:.1f} us of host time per eager step (wall incl. GPU {1e6 * (time.perf_counter() - t0) / 300:.1f})")
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
