"""One case of tools/r6/fuzz_rowpos.py again, row by row: the HIP gradients, the C oracle's and float32 / float64 autograd of the torch restatement of the reference.
    python tools/r6/fuzz_rowpos_repro.py <seed0> <cases to scan> [how many failing cases to print = 2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "r6"))
import numpy as np
import torch
import fuzz_rowpos as fz
from sot_amd import _native as nat
from sot_amd import _torch_path as tpath
from oracle import sot_oracle as so

seed0, n_cases = int(sys.argv[1]), int(sys.argv[2])
n_print = int(sys.argv[3]) if len(sys.argv) > 3 else 2
recorded = []
orig = so.backward
def spy(x, y, xp, yp, gr, p, flags):
    out = orig(x, y, xp, yp, gr, p=p, flags=flags)
    recorded.append(dict(x=x, y=y, xp=xp, yp=yp, gr=gr, p=p, flags=flags, wx=out[0], wy=out[1]))
    return out
so.backward = spy
fz.run(budget=600, seed0=seed0, max_cases=n_cases, verbose=False)
dev = torch.device("cuda:0")
printed = 0
for c in recorded:
    if printed >= n_print:
        break
    x, y, xp, yp = (torch.as_tensor(c[k]) for k in ("x", "y", "xp", "yp"))
    fl = int(c["flags"]) | nat.FLAG_REQUIRE_SORT
    gx, gy = nat.backward_rows(x.to(dev), y.to(dev), xp.to(dev), yp.to(dev), c["p"], fl, torch.as_tensor(c["gr"]).float().to(dev), grad_scale=1.0)
    gx, gy = gx.cpu().numpy(), gy.cpu().numpy()
    scale = np.maximum(np.abs(c["wx"]).max(1), np.abs(c["wy"]).max(1))[:, None] + 1e-30
    if max(np.abs(gx - c["wx"]).max(1).max() / 1, 0) == 0 or (np.maximum(np.abs(gx - c["wx"]).max(1), np.abs(gy - c["wy"]).max(1)) / scale[:, 0]).max() <= 2e-4:
        continue
    printed += 1
    kw = dict(p=c["p"], square_dist=bool(fl & 1), dont_normalize=bool(fl & 2), limit_quantile_range=bool(fl & 4), require_sort=True, hinge_on=False)
    res = {}
    for dt in (torch.float32, torch.float64):
        xx, yy = x.to(dt).clone().requires_grad_(True), y.to(dt).clone().requires_grad_(True)
        rows = tpath.module_forward(xx, yy, xp.to(dt), yp.to(dt), rows_only=True, **kw)
        (rows * torch.as_tensor(c["gr"]).to(dt)).sum().backward()
        res[dt] = (xx.grad.numpy(), yy.grad.numpy(), rows.detach().numpy())
    hip_rows = nat.forward_rows(x.to(dev), y.to(dev), xp.to(dev), yp.to(dev), c["p"], fl).cpu().numpy()
    ora_rows = so.forward(c["x"], c["y"], c["xp"], c["yp"], p=c["p"], flags=fl & 15)
    print("p", c["p"], "flags", fl, "shape", x.shape)
    for r in range(x.shape[0]):
        e_or = max(np.abs(gx[r] - c["wx"][r]).max(), np.abs(gy[r] - c["wy"][r]).max()) / scale[r, 0]
        e_32 = max(np.abs(gx[r] - res[torch.float32][0][r]).max(), np.abs(gy[r] - res[torch.float32][1][r]).max()) / scale[r, 0]
        e_64 = max(np.abs(gx[r] - res[torch.float64][0][r]).max(), np.abs(gy[r] - res[torch.float64][1][r]).max()) / scale[r, 0]
        o_64 = max(np.abs(c["wx"][r] - res[torch.float64][0][r]).max(), np.abs(c["wy"][r] - res[torch.float64][1][r]).max()) / scale[r, 0]
        t_64 = max(np.abs(res[torch.float32][0][r] - res[torch.float64][0][r]).max(), np.abs(res[torch.float32][1][r] - res[torch.float64][1][r]).max()) / scale[r, 0]
        nx, ny = len(np.unique(c["xp"][r])), len(np.unique(c["yp"][r]))
        if e_or > 2e-5:
            t32x, t32y = res[torch.float32][0][r], res[torch.float32][1][r]
            for nm, hg, tg, w, pos in (("x", gx[r], t32x, c["x"][r], c["xp"][r]), ("y", gy[r], t32y, c["y"][r], c["yp"][r])):
                d = np.abs(hg - tg)
                top = np.argsort(-d)[:4]
                srt = np.argsort(pos, kind="stable")
                rank = np.empty_like(srt); rank[srt] = np.arange(len(srt))
                print(f"    {nm}: sum w {w.sum():.6g} max w {w.max():.4g}; largest |HIP - torch32| at", [(int(i), f"hip {hg[i]:.3g}", f"torch {tg[i]:.3g}", f"w {w[i]:.3g}", f"pos {pos[i]:.6g}", f"rank {int(rank[i])}", f"dups {int((pos == pos[i]).sum())}") for i in top])
            print(f"    rows: HIP {hip_rows[r]:.6g} oracle {ora_rows[r]:.6g} torch32 {res[torch.float32][2][r]:.6g} torch64 {res[torch.float64][2][r]:.6g}; max|g|: HIP {np.abs(gx[r]).max():.3g} oracle {np.abs(c['wx'][r]).max():.3g} torch32 {np.abs(res[torch.float32][0][r]).max():.3g} torch64 {np.abs(res[torch.float64][0][r]).max():.3g}; gr {c['gr'][r]:.3g}")
            print(f"row {r:3d}: HIP-oracle {e_or:.2e}  HIP-torch32 {e_32:.2e}  HIP-torch64 {e_64:.2e}  oracle-torch64 {o_64:.2e}  torch32-torch64 {t_64:.2e}   distinct positions {nx}/{ny}")
    