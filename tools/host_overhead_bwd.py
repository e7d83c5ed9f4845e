import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sot_amd.losses import Wasserstein1D
dev = torch.device("cuda:0")
B, N = 64, 1025
x = torch.rand(B, N, device=dev)
yg = torch.rand(B, N, device=dev).requires_grad_(True)
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
def f():
    yg.grad = None
    mod(x, yg, x_pos=pos, y_pos=pos2).backward()
for _ in range(100): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000): f()
print("fwd+bwd host us/call", (time.perf_counter() - t0) / 1000 * 1e6)
# fwd only with grad
t0 = time.perf_counter()
for _ in range(1000): l = mod(x, yg, x_pos=pos, y_pos=pos2)
print("fwd(with graph) host us/call", (time.perf_counter() - t0) / 1000 * 1e6)
# reference point: a trivial torch op fwd+bwd
def g():
    yg.grad = None
    (yg * 2.0).sum().backward()
for _ in range(100): g()
t0 = time.perf_counter()
for _ in range(1000): g()
print("trivial torch fwd+bwd host us/call", (time.perf_counter() - t0) / 1000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): f()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
