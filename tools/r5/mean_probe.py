"""The training form at the paper's step (1024 x 1025, paper mode): separate batch-mean kernel vs the mean by the row kernel's last workgroup,
each replayed from a HIP graph: python3 tools/r5/mean_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sot_amd import _native as nat
from sot_amd import spectra
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for rows in (1024, 4096):
    x, y = torch.rand(rows, 1025, device=dev, generator=g), torch.rand(rows, 1025, device=dev, generator=g)
    pos = spectra.unit_frequencies(2048, 16000.0, dev)
    plan = nat.PositionPlan(pos, pos.clone())
    for fused in (False, True):
        fn = lambda: nat.loss_and_grad(x, y, pos, pos, 2.0, 15, plan, fused_mean=fused)
        for _ in range(5):
            fn()
        side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20): gr.replay()
        e0.record()
        for _ in range(200): gr.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{rows} x 1025 training form, mean {'in the row kernel' if fused else 'as its own kernel'}: {1e3 * e0.elapsed_time(e1) / 200:.2f} us per replay, loss {float(out[0]):.9g}")
