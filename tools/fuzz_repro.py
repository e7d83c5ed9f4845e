"""Details of one tools/fuzz_gpu.py case: python tools/fuzz_repro.py "<the dict it printed>" -- the specialised kernels, the generic kernels
(FLAG_NO_SPECIALIZE) and the C oracle side by side, row by row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sot_amd import _native as nat
from oracle import sot_oracle as so
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_gpu as fz
positions, weights, dev = fz.positions, fz.weights, fz.dev
np.set_printoptions(precision=6, linewidth=200)
for arg in sys.argv[1:]:
    d = eval(arg)
    g = torch.Generator().manual_seed(d["seed"])
    n, m, B, pk = d["n"], d["m"], d["B"], d["pos"]
    xpos = positions(pk, n, B, g)
    ypos = xpos.clone() if d["same"] else positions(pk, m, B, g)
    x, y = weights(d["w"], B, n, m, d["seed"])
    xd, yd, xpd, ypd = x.to(dev), y.to(dev), xpos.to(dev), ypos.to(dev)
    flags, p = d["flags"], d["p"]
    plan = nat.PositionPlan(xpd, ypd) if d["plan"] else None
    grow = torch.linspace(0.5, 1.5, B)
    gx, gy = nat.backward_rows(xd, yd, xpd, ypd, p, flags, grow.to(dev), plan=plan, grad_scale=0.5)
    hx, hy = nat.backward_rows(xd, yd, xpd, ypd, p, flags | nat.FLAG_NO_SPECIALIZE, grow.to(dev), plan=plan, grad_scale=0.5)
    wx, wy = so.backward(x.numpy(), y.numpy(), xpos.numpy(), ypos.numpy(), (0.5 * grow).numpy(), p=p, flags=flags & 15)
    # float64 autograd of the package's torch-op route on the CPU: the arithmetic truth where no levels tie
    from sot_amd import _torch_path as tp
    x64, y64 = x.double().requires_grad_(True), y.double().requires_grad_(True)
    rows64 = tp.module_forward(x64, y64, xpos.double(), ypos.double(), p=p, square_dist=bool(flags & 1), dont_normalize=bool(flags & 2),
                               limit_quantile_range=bool(flags & 4), require_sort=True, hinge_on=False, rows_only=True)
    (rows64 * (0.5 * grow).double()).sum().backward()
    tx, ty = x64.grad.numpy(), y64.grad.numpy()
    print("==", d)
    for name, gg, ww, tt in (("gx", gx.cpu().numpy(), wx, tx), ("gy", gy.cpu().numpy(), wy, ty)):
        sc = max(np.abs(tx).max(), np.abs(ty).max())
        print(f"   {name}: |hip - f64| / scale {np.abs(gg - tt).max() / sc:.3g}   |oracle - f64| / scale {np.abs(ww - tt).max() / sc:.3g}   (whole batch, scale {sc:.3g})")
    print("specialised == generic:", bool(torch.equal(gx, hx)), bool(torch.equal(gy, hy)))
    for name, gg, ww, w_in in (("gx", gx.cpu().numpy(), wx, x.numpy()), ("gy", gy.cpu().numpy(), wy, y.numpy())):
        scale = np.abs(ww).max(axis=1, keepdims=True) + 1e-30
        e = np.abs(gg - ww) / scale
        rows = np.argsort(-e.max(axis=1))[:2]
        for r in rows:
            if e[r].max() <= 2e-5:
                continue
            k = int(np.argmax(e[r]))
            lo, hi = max(0, k - 3), min(gg.shape[1], k + 4)
            print(f" {name} row {r} (mass x {x[r].sum():.4g}, y {y[r].sum():.4g}; row max |want| {scale[r, 0]:.4g}) worst at {k}: got {gg[r, lo:hi]} want {ww[r, lo:hi]} w {w_in[r, lo:hi]}")
            if n <= 8 and m <= 8:
                print("   x", x[r].numpy(), "y", y[r].numpy(), "xpos", xpos.numpy() if xpos.ndim == 1 else xpos[r].numpy())
                print("   got gx", gx[r].cpu().numpy(), "gy", gy[r].cpu().numpy()); print("   want gx", wx[r], "gy", wy[r])
