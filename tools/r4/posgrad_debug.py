import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import torch_restatement as tr
from sot_amd import losses as L
dev = torch.device("cuda:0")
B, n, m = 5, 2048, 2048
mode, shared = "p2_cutoff_dn", False
g = torch.Generator().manual_seed(1000 * n + m + len(mode) + int(shared))
kw = dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
x, y = torch.rand(B, n, generator=g) + 0.01, torch.rand(B, m, generator=g) + 0.01
def positions(width):
    pos = torch.rand(1 if shared else B, width, generator=g)
    pos = torch.sort(pos, dim=1)[0]
    return pos[0].clone() if shared else pos
xp, yp = positions(n), positions(m)
up = torch.rand(B, generator=g) + 0.5
xpr, ypr = xp.clone().requires_grad_(True), yp.clone().requires_grad_(True)
(tr.sot_loss(x, y, xpr, ypr, reduce=False, **kw) * up).sum().backward()
mod = L.Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
xpd, ypd = xp.to(dev).requires_grad_(True), yp.to(dev).requires_grad_(True)
(mod.row_losses(x.to(dev), y.to(dev), xpd, ypd) * up.to(dev)).sum().backward()
for name, got, want in (("x", xpd.grad.cpu(), xpr.grad), ("y", ypd.grad.cpu(), ypr.grad)):
    err = (got - want).abs()
    bad = (err > 2e-6 * want.abs().max()).nonzero()
    print(name, "max err", float(err.max()), "scale", float(want.abs().max()), "bad entries", bad.tolist()[:20])
    for r, c in bad.tolist()[:6]:
        print("  row", r, "col", c, "got", got[r, max(c-2,0):c+3].tolist(), "want", want[r, max(c-2,0):c+3].tolist())
# the same through float64 autograd
x64, y64 = x.double(), y.double()
xp64, yp64 = xp.double().requires_grad_(True), yp.double().requires_grad_(True)
(tr.sot_loss(x64, y64, xp64, yp64, reduce=False, **kw) * up.double()).sum().backward()
print("f64 vs f32 ref x", float((xp64.grad.float() - xpr.grad).abs().max()), "hip vs f64 x", float((xpd.grad.cpu() - xp64.grad.float()).abs().max()))
print("f64 vs f32 ref y", float((yp64.grad.float() - ypr.grad).abs().max()), "hip vs f64 y", float((ypd.grad.cpu() - yp64.grad.float()).abs().max()))
uq, vq, lv, cu, cv = tr.sot_loss(x, y, xp, yp, return_quantiles=True, **kw)
for r in range(B):
    ties = (lv[r, 1:] == lv[r, :-1]).sum()
    print("row", r, "tied levels", int(ties), "levels > 1:", int((lv[r] > 1).sum()), "cu last", float(cu[r, -1]), "cv last", float(cv[r, -1]))
