import ctypes, os, sys
ROOT='/root/repo'; sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "tools", "ablate_libs", "stamps.so")
os.environ["SOT_LIB_PATH"] = lib
import torch
from sot_amd import _native as nat
from sot_amd.bench_inputs import ragged_supports
from sot_amd.losses import wasserstein_1d_csr
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
rs = ragged_supports(8192, 512, 1234)
(xw, xp, xo), (yw, yp, yo) = [[t.to(dev) for t in part] for part in rs["csr"]]
kw = dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
for _ in range(20): wasserstein_1d_csr(xw, xp, xo, yw, yp, yo, rs["max_n"], rs["max_m"], **kw)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
ctypes.CDLL(lib).sot_debug_read_stamps(out, 64)
names = ["row start", "staged", "P2 chunk sums", "P3 fold", "P3b division/gather", "CDFs built", "partition search", "walk", "row done"]
prev = out[0]
for i, n in enumerate(names):
    if out[i]:
        print(f"  {n:28s} +{out[i] - prev:7d} cycles"); prev = out[i]
wg = [out[32 + i] for i in range(16)]
print("rows done at", [w - wg[0] for w in wg[2:10] if w])
