"""Round 6: per-row positions, unsorted rows -- pre-sort kernel + gathering row kernel (default) against the row kernel's own in-LDS merge
sort (SOT_FLAG_NO_SPECIALIZE keeps the pre-sort off), by row length: where the split pays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for B, N in ((8192, 256), (8192, 512), (4096, 768), (4096, 1024), (4096, 1025), (4096, 1536), (4096, 2000), (4096, 2048)):
    x, y = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
    px, py = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
    a = nat.forward_rows(x, y, px, py, 2.0, 15, None)
    b = nat.forward_rows(x, y, px, py, 2.0, 15 | 32, None)
    t1 = timed(lambda: nat.forward_rows(x, y, px, py, 2.0, 15, None))
    t0 = timed(lambda: nat.forward_rows(x, y, px, py, 2.0, 15 | 32, None))
    print(f"{B} x {N}: pre-sort + gather {t1:7.1f} us   in-kernel merge sort {t0:7.1f} us   equal {torch.equal(a, b)}")
