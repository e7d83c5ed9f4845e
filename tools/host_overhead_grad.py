import cProfile, pstats, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sot_amd.losses import Wasserstein1D
dev = torch.device("cuda:0")
B, N = 64, 1025
x, y = torch.rand(B, N, device=dev), torch.rand(B, N, device=dev)
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
yg = y.clone().requires_grad_(True)
def f_grad():
    yg.grad = None
    mod(x, yg, x_pos=pos, y_pos=pos2).backward()
def f_fwd_only():
    return mod(x, yg, x_pos=pos, y_pos=pos2)
def f_trivial():
    yg.grad = None
    (yg * 2.0).sum().backward()
for f in (f_grad, f_fwd_only, f_trivial):
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): f()
    print(f.__name__, (time.perf_counter() - t0) / 1000 * 1e6, "us host")
    torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): f_grad()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(18)
