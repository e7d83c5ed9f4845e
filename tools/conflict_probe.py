"""How sensitive is the forward kernel to LDS bank conflicts in the merge?  Same work, different regularity."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sot_amd.losses import Wasserstein1D
dev = torch.device("cuda:0")
B, N = 8192, 2048
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
mod = Wasserstein1D(p=1).to(dev)
def run(name, mk):
    sets = [mk(i) for i in range(6)]
    for i in range(5): mod(*sets[i % 6], x_pos=pos, y_pos=pos2)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    with torch.no_grad():
        for i in range(60): mod(*sets[i % 6], x_pos=pos, y_pos=pos2)
    b.record(); torch.cuda.synchronize()
    print(f"{name:46s} {a.elapsed_time(b) / 60 * 1e3:7.1f} us/call")
g = torch.Generator(device=dev).manual_seed(0)
def rnd(i): return (torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g))
def same(i):
    x = torch.rand(B, N, device=dev, generator=g); return (x, x.clone())
def const(i):
    x = torch.ones(B, N, device=dev) + 1e-3 * torch.rand(B, N, device=dev, generator=g); return (x, x.flip(1).contiguous())
def peaky(i): return (torch.rand(B, N, device=dev, generator=g) ** 8, torch.rand(B, N, device=dev, generator=g) ** 8)
run("uniform random x, y (benchmark data)", rnd)
run("y == x (perfectly alternating merge)", same)
run("near-constant rows (regular stride, U~V)", const)
run("peaky U^8 (long tie runs, clustered merge)", peaky)
