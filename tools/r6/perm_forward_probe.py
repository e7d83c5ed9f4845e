"""Round 6: what a per-row-position forward costs when the rows' sort permutations are HANDED to it (row_perm_in: it gathers, it does not sort) --
the second half of a 'sort kernel + row kernel' split, measured before building the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
B, N = 4096, 2048
x, y = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
px, py = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


perm = nat.row_permutations(x, y, px, py, 15)
ref = nat.forward_rows(x, y, px, py, 2.0, 15, None, perm_out=perm).clone()
got = nat.forward_rows(x, y, px, py, 2.0, 15, None, perm_in=perm)
print("perm_in forward == sorting forward:", torch.equal(ref, got))
print("forward, sorting: %.1f us" % timed(lambda: nat.forward_rows(x, y, px, py, 2.0, 15, None)))
print("forward, sorting + storing the permutations: %.1f us" % timed(lambda: nat.forward_rows(x, y, px, py, 2.0, 15, None, perm_out=perm)))
print("forward, permutations handed over: %.1f us" % timed(lambda: nat.forward_rows(x, y, px, py, 2.0, 15, None, perm_in=perm)))
sx, sy = torch.sort(px, 1).values, torch.sort(py, 1).values
print("forward, rows already sorted: %.1f us" % timed(lambda: nat.forward_rows(x, y, sx, sy, 2.0, 15, None)))
