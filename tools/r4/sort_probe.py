"""sot_segmented_sort 4096 x 2048 in a loop (for rocprofv3 PMC passes) + its stream time and torch.sort's on the same keys."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 2048)
g = torch.Generator(device=dev).manual_seed(1)
keys = torch.rand(B, N, device=dev, generator=g)
for _ in range(3):
    v, i = nat.segmented_sort(keys)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    v, i = nat.segmented_sort(keys)
e1.record(); torch.cuda.synchronize()
t_ours = e0.elapsed_time(e1) / 20
e0.record()
for _ in range(20):
    tv, ti = torch.sort(keys, dim=1, stable=True)
e1.record(); torch.cuda.synchronize()
print(f"sot_segmented_sort {B}x{N}: {1e3*t_ours:.1f} us; torch.sort(stable): {1e3*e0.elapsed_time(e1)/20:.1f} us; equal: {torch.equal(v, tv) and torch.equal(i, ti)}")
