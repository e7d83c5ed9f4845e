"""BASELINE config 4: B=8192, N=512 peaky spectra with a per-row amplitude cutoff -> ragged supports.
Times the masked-dense form (zeros kept; dense kernel) and the CSR form (sot_w1d_forward_csr) with HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle.inputs import gen_inputs
from sot_amd import _native as nat
from sot_amd.losses import wasserstein_1d_csr

dev = torch.device("cuda:0")
B, N = 8192, 512
x, y = gen_inputs("peaky", B, N, N, 1234)
g = torch.Generator().manual_seed(1234)
tau = 10 ** (-3 + 2.7 * torch.rand(B, 1, generator=g))
kx, ky = x >= tau * x.amax(1, keepdim=True), y >= tau * y.amax(1, keepdim=True)
xm, ym = torch.where(kx, x, torch.zeros_like(x)).to(dev), torch.where(ky, y, torch.zeros_like(y)).to(dev)
pos = torch.linspace(0, 1, N)


def to_csr(dense, keep):
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(keep.sum(1), 0)
    return dense[keep].to(dev), pos.expand_as(dense)[keep].to(dev), off.to(dev)


xw, xp, xo = to_csr(x, kx)
yw, yp, yo = to_csr(y, ky)
max_n, max_m = int(kx.sum(1).max()), int(ky.sum(1).max())
posd = pos.to(dev); posd2 = posd.clone()
kept = float(kx.sum() + ky.sum()) / (2 * B)


def timed(fn, n=200, warm=300):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, flags, p in (("p1", 8, 1.0), ("paper cutoff", 15, 2.0)):
    kw = dict(p=p, square_dist=bool(flags & 1), dont_normalize=bool(flags & 2), limit_quantile_range=bool(flags & 4))
    t_dense = timed(lambda: nat.forward_rows(xm, ym, posd, posd2, p, flags))
    t_csr = timed(lambda: wasserstein_1d_csr(xw, xp, xo, yw, yp, yo, max_n, max_m, **kw))
    dense_bytes = B * (8 * N + 4)
    csr_bytes = 8 * (xw.numel() + yw.numel()) + 16 * (B + 1) + 4 * B
    print(f"{name:13s} mean kept support {kept:5.1f} of {N} (max {max_n}/{max_m}): masked-dense {t_dense:6.1f} us ({B / t_dense:6.1f} Mrows/s, "
          f"{dense_bytes / t_dense / 1e3:5.0f} GB/s)   CSR {t_csr:6.1f} us ({B / t_csr:6.1f} Mrows/s, {csr_bytes / t_csr / 1e3:5.0f} GB/s)")
