"""Per-row positions through the generic row kernels (4096 x 2048, paper mode): forward (unsorted / sorted rows), backward, position gradients; and the segmented sort."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
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
print("per-row positions forward 4096x2048 (paper mode): %.1f us" % timed(lambda: nat.forward_rows(x, y, px, py, 2.0, 15, None)))
sx, sy = torch.sort(px, 1).values, torch.sort(py, 1).values
print("  rows already sorted: %.1f us" % timed(lambda: nat.forward_rows(x, y, sx, sy, 2.0, 15, None)))
one = torch.ones(1, device=dev)
for label, fn in (("per-row positions backward (both gradients)", lambda: nat.backward_rows(x, y, px, py, 2.0, 15, one)),
                  ("per-row position gradients", lambda: nat.position_grads(x, y, px, py, 2.0, 15, one)),
                  ("sot_segmented_sort 4096x2048", lambda: nat.segmented_sort(px))):
    try:
        print("%s: %.1f us" % (label, timed(fn)))
    except Exception as exc:   # diagnostic library variants hold a subset of the kernels
        print("%s: not in this library (%s)" % (label, type(exc).__name__))
x5, y5, p5a, p5b = x[:, :512].contiguous(), y[:, :512].contiguous(), px[:, :512].contiguous(), py[:, :512].contiguous()
print("per-row positions forward 4096x512: %.1f us" % timed(lambda: nat.forward_rows(x5, y5, p5a, p5b, 2.0, 15, None)))
