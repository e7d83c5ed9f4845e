"""Round 6: the one-wavefront-per-row segmented sort (csrc/sot_wave_sort.hpp) against torch.sort(stable=True), values and indices bit-exact:
every row length class, generic / tied / clustered / adversarial keys (all inside one quantisation bin: the merge-sort fallback), NaN, +-inf,
+-0; then its stream time at 4096 x 2048 beside torch.sort's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(6)
bad = 0


def check(name, keys):
    global bad
    v, i = nat.segmented_sort(keys.to(dev))
    tv, ti = torch.sort(keys, dim=1, stable=True)
    ok = torch.equal(v.cpu(), tv) and torch.equal(i.cpu(), ti)
    if not ok:
        bad += 1
        nv = (v.cpu() != tv).sum().item(); ni = (i.cpu() != ti).sum().item()
        print(f"MISMATCH {name} shape {tuple(keys.shape)}: {nv} values, {ni} indices differ")
    return ok


for n in (1, 2, 63, 64, 65, 100, 128, 129, 257, 500, 512, 513, 777, 1024, 1025, 1500, 2047, 2048):
    B = 37
    u = torch.rand(B, n, generator=g)
    check("uniform", u)
    check("normal", torch.randn(B, n, generator=g) * 3)
    check("ties", torch.round(u * 50) / 50)
    check("descending", torch.sort(u, dim=1, descending=True)[0].contiguous())
    check("one bin", 0.5 + u * 2.0 ** -12)                       # all keys inside one 2^-12 interval: the adaptive range still separates them
    check("cluster + outlier", torch.cat([u[:, :-1] * 1e-9, torch.ones(B, 1)], 1) if n > 1 else u)   # everything but one key in ONE bin: fallback
    check("all equal", torch.full((B, n), 0.25))
    z = u.clone(); z[:, ::3] = 0.0; z[:, 1::5] = -0.0; check("+-0", z - 0.0 * z)
    w = u.clone(); w[:, 0] = float("inf"); check("+inf", w)
    w = u.clone(); w[:, -1] = float("-inf"); check("-inf", w)
    w = (u - 0.5) * 3e38 * 2; check("huge range", w)
    w = u * 1e-42; check("denormals", w)
    if n > 2:
        w = u.clone(); w[::2, 1] = float("nan")
        v, i = nat.segmented_sort(w.to(dev))      # NaN keys: no defined order here; indices must stay inside the row and rows without NaN must be exact
        assert int(i.min()) >= 0 and int(i.max()) < n
        tv, ti = torch.sort(w[1::2], dim=1, stable=True)
        if not (torch.equal(v.cpu()[1::2], tv) and torch.equal(i.cpu()[1::2], ti)):
            bad += 1; print("MISMATCH NaN-free rows beside NaN rows", n)
print("cases with mismatches:", bad)

B, N = 4096, 2048
gd = torch.Generator(device=dev).manual_seed(1)
keys = torch.rand(B, N, device=dev, generator=gd)
for _ in range(3):
    v, i = nat.segmented_sort(keys)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(20):
        v, i = nat.segmented_sort(keys)
    e1.record(); torch.cuda.synchronize()
    t_ours = e0.elapsed_time(e1) / 20
    print(f"sot_segmented_sort {B}x{N}: {1e3 * t_ours:.1f} us")
e0.record()
for _ in range(20):
    tv, ti = torch.sort(keys, dim=1, stable=True)
e1.record(); torch.cuda.synchronize()
print(f"torch.sort(stable): {1e3 * e0.elapsed_time(e1) / 20:.1f} us; equal: {torch.equal(v, tv) and torch.equal(i, ti)}")
for nn in (512, 1024):
    k2 = torch.rand(8192, nn, device=dev, generator=gd)
    for _ in range(3):
        nat.segmented_sort(k2)
    e0.record()
    for _ in range(20):
        nat.segmented_sort(k2)
    e1.record(); torch.cuda.synchronize()
    print(f"sot_segmented_sort 8192x{nn}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us")
