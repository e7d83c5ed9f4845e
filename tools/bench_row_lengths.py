"""Every row length with a compile-time kernel against the generic kernels (SOT_FLAG_NO_SPECIALIZE): forward and backward
w.r.t. y, p = 1 and the paper's cutoff mode, back-to-back launches timed with HIP events.  usage: python tools/bench_row_lengths.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from sot_amd import _native as nat
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N, B in ((1024, 16384), (4096, 4096), (129, 65536), (2049, 8192), (2048, 8192), (1025, 16384)):
    x = torch.rand(B, N, device=dev); y = torch.rand(B, N, device=dev)
    pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2)
    g = torch.ones(B, device=dev)
    for flags, p, name in ((8, 1.0, "p1"), (1 | 2 | 4 | 8, 2.0, "paper")):
        f_spec = t(lambda: nat.forward_rows(x, y, pos, pos2, p, flags, plan))
        f_gen = t(lambda: nat.forward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, plan))
        b_spec = t(lambda: nat.backward_rows(x, y, pos, pos2, p, flags, g, need_gx=False, plan=plan))
        b_gen = t(lambda: nat.backward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, g, need_gx=False, plan=plan))
        print(f"{B}x{N} {name}: forward {f_spec:.1f} us (generic {f_gen:.1f}) | backward(y) {b_spec:.1f} us (generic {b_gen:.1f})", flush=True)
