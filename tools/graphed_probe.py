"""The drop-in module under torch.cuda.make_graphed_callables at the paper's step (1024 rows x 1025 bins, paper mode): host time per
forward + backward and equality with the eager module."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sot_amd.losses import Wasserstein1D
from sot_amd import spectra

dev = torch.device("cuda:0")
B, N = 1024, 1025
mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True, fixed_x=N).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, N, device=dev, generator=g)
y = torch.rand(B, N, device=dev, generator=g).requires_grad_(True)

def eager():
    y.grad = None
    loss = mod(x, y)
    loss.backward()
    return loss.detach().clone(), y.grad.clone()

l0, g0 = eager()
xs = x.clone()
ys = torch.rand(B, N, device=dev, generator=g).requires_grad_(True)
graphed = torch.cuda.make_graphed_callables(mod, (xs, ys))
ys2 = y.detach().clone().requires_grad_(True)
loss = graphed(x, ys2)
loss.backward()
torch.cuda.synchronize()
print("graphed == eager: loss", bool(torch.equal(loss.detach(), l0)), "grad", bool(torch.equal(ys2.grad, g0)))

def timeit(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return host * 1e6, (time.perf_counter() - t) / n * 1e6

def step_eager():
    y.grad = None
    mod(x, y).backward()

def step_graphed():
    ys2.grad = None
    graphed(x, ys2).backward()

for name, fn in (("eager", step_eager), ("graphed", step_graphed), ("eager", step_eager), ("graphed", step_graphed)):
    h, w = timeit(fn)
    print(f"{name:8s} host {h:6.1f} us per step, wall {w:6.1f} us per step")
