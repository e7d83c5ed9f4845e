"""Round 6: time of the per-row-position forward (4096 x 2048, unsorted rows: pre-sort kernel + gathering row kernel) -- run under
rocprofv3 --kernel-trace --stats for the split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 2048)
x, y = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
px, py = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
for _ in range(5):
    nat.forward_rows(x, y, px, py, 2.0, 15, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    nat.forward_rows(x, y, px, py, 2.0, 15, None)
e1.record(); torch.cuda.synchronize()
print(f"per-row positions forward {B}x{N}: {1e3 * e0.elapsed_time(e1) / 30:.1f} us")
