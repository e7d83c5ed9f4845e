"""Host-side cost per call of the Python boundary (tiny batch so that GPU time is negligible)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sot_amd import _native as nat  # noqa: E402
from sot_amd.losses import Wasserstein1D  # noqa: E402

dev = torch.device("cuda:0")
B, N = 64, 1025
x, y = torch.rand(B, N, device=dev), torch.rand(B, N, device=dev)
pos = torch.linspace(0, 1, N, device=dev)
pos2 = pos.clone()
mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
yg = y.clone().requires_grad_(True)


def bench(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6


def f_nograd():
    with torch.no_grad():
        return mod(x, y, x_pos=pos, y_pos=pos2)


def f_grad():
    yg.grad = None
    mod(x, yg, x_pos=pos, y_pos=pos2).backward()


def f_raw():
    return nat.forward_rows(x, y, pos, pos2, 2.0, 15)


print("module fwd (no_grad): host %.1f us/call, incl. drain %.1f" % bench(f_nograd))
print("raw forward_rows    : host %.1f us/call, incl. drain %.1f" % bench(f_raw))
print("module fwd+bwd      : host %.1f us/call, incl. drain %.1f" % bench(f_grad, 1000))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    f_nograd()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
