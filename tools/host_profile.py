"""Where the host time of one training-style call of the module goes (cProfile over 3000 forward+backward calls, 8192 x 2048 paper mode).
Usage: python tools/host_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sot_amd.losses import Wasserstein1D
dev = torch.device("cuda:0")
B, N = 8192, 2048
x = torch.rand(B, N, device=dev)
y = torch.rand(B, N, device=dev, requires_grad=True)
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)


def step():
    y.grad = None
    mod(x, y, x_pos=pos, y_pos=pos2).backward()


for _ in range(50):
    step()
torch.cuda.synchronize()
# host cost alone: a tiny problem keeps the GPU ahead of the host
xs, ys = x[:8].contiguous(), y[:8].detach().clone().requires_grad_(True)


def small():
    ys.grad = None
    mod(xs, ys, x_pos=pos, y_pos=pos2).backward()


for _ in range(50):
    small()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    small()
torch.cuda.synchronize()
print("host-bound time per forward+backward call: %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    small()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
