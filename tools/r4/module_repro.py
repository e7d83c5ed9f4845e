"""Reproduce one tools/fuzz_module.py case (rep 0) from its seed: GPU module vs CPU float32 vs CPU float64 gradients, and the tied CDF levels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sot_amd.losses import Wasserstein1D
from sot_amd import _torch_path as tp
dev = torch.device("cuda:0")
seed, N, lead = 2093900612, 64, (34,)
ctor = dict(p=2, square_dist=False, dont_normalize=False, limit_quantile_range=False, hinge=True)
kwargs = dict(hinge=0.0, dims=0)
g = torch.Generator().manual_seed(seed)
pos_args = dict(x_pos=torch.rand(N, generator=g), y_pos=torch.rand(N, generator=g))
mk = lambda: 0.05 + torch.rand(lead + (N,), generator=g) ** 3
x, y = mk(), mk()
w = torch.rand((), generator=g) if False else None
res = {}
for name, dt, dv in (("cpu32", torch.float32, "cpu"), ("cpu64", torch.float64, "cpu"), ("gpu", torch.float32, dev)):
    mod = Wasserstein1D(**ctor).to(dv)
    yy = y.to(dt).to(dv).requires_grad_(True)
    out = mod(x.to(dt).to(dv), yy, **{k: v.to(dt).to(dv) for k, v in pos_args.items()}, **kwargs)
    if w is None:
        w = torch.rand(out.shape, generator=g)
    (out * w.to(dt).to(dv)).sum().backward()
    res[name] = (out.detach().cpu().double(), yy.grad.cpu().double())
    print(name, "out", out.detach().cpu().flatten()[:4].tolist())
for a, b in (("gpu", "cpu32"), ("gpu", "cpu64"), ("cpu32", "cpu64")):
    d = (res[a][1] - res[b][1]).abs()
    r, c = divmod(int(d.argmax()), N)
    print(f"{a} vs {b}: max grad diff {float(d.max()):.3e} / max {float(res[b][1].abs().max()):.3e} at row {r} col {c}: {float(res[a][1][r, c]):.6e} vs {float(res[b][1][r, c]):.6e}")
# ties in the float32 CDFs of the worst row
uq, vq, lv, cu, cv = tp.module_forward(x, y, pos_args["x_pos"], pos_args["y_pos"], p=2, square_dist=False, dont_normalize=False, limit_quantile_range=False,
                                       require_sort=True, hinge_on=False, return_quantiles=True)
d = (res["gpu"][1] - res["cpu32"][1]).abs()
r = int(d.max(1).values.argmax())
print("row", r, "tied merged levels:", int((lv[r, 1:] == lv[r, :-1]).sum()), "row grad diffs > 1e-3 max:", (d[r] > 1e-3 * res["cpu32"][1].abs().max()).nonzero().flatten().tolist())
print("x_pos ties:", int(N - pos_args["x_pos"].unique().numel()), "y_pos ties:", int(N - pos_args["y_pos"].unique().numel()))
