"""Interleaved A/B of the batch mean: in the row kernel's last workgroup (one kernel) vs the separate mean kernel (two),
through the same FFI call (sot_w1d_loss).  B=8192 x N (default 2048), six rotating input sets, one stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sot_amd import _native as nat

nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
B, N = int(os.environ.get("AB_B", "8192")), int(os.environ.get("AB_N", "2048"))
flags, p = int(os.environ.get("AB_FLAGS", "8")), float(os.environ.get("AB_P", "1.0"))
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(6)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
rows, mean = torch.empty(B, device=dev), torch.empty(1, device=dev)


def run(fused, n):
    for i in range(n):
        nat.loss_fused(*sets[i % 6], pos, pos2, p, flags, plan, fused_mean=fused, row_out=rows, mean_out=mean)


def fwd_only(n):
    for i in range(n):
        nat.forward_rows(*sets[i % 6], pos, pos2, p, flags, plan, rows)


run(True, 300); run(False, 300)
for rnd in range(5):
    res = []
    for name, fn in (("one kernel", lambda n: run(True, n)), ("two kernels", lambda n: run(False, n)), ("rows only", fwd_only)):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(300); b.record(); torch.cuda.synchronize()
        res.append(f"{name} {a.elapsed_time(b) / 300 * 1e3:6.2f} us")
    print(" | ".join(res))
