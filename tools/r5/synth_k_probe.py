"""Synthesiser forward / forward + backward by the number of partials K (256 clips x 16 frames -> 4096 samples).  K = 8 (the paper's data:
`40_1950_4096_04_1_4000_8_1_harmonic`) puts the tile kernels' LDS rows at a 64-dword stride (8-way bank conflicts: 66 % of their LDS cycles
in the PMC passes) -- and is still the fastest per partial (6.7 / 15.2 us against 7.3 / 18.0 at K = 7, 8.2 / 18.4 at K = 9): the kernels are
bound by `sincosf` and the float64 phase sums, not by LDS; a padded row stride was therefore not built.  python3 tools/r5/synth_k_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sot_amd import spectra
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
def timed(fn, reps=100):
    best = 1e9
    for _ in range(4):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best
for K in (7, 8, 9, 16, 17):
    amp = torch.rand(256, 16, K, device=dev, generator=gen).requires_grad_(True)
    f0 = (40 + 1900 * torch.rand(256, 16, 1, device=dev, generator=gen)).requires_grad_(True)
    g = torch.randn(256, 4096, device=dev, generator=gen)
    with torch.no_grad():
        fwd = timed(lambda: spectra.sinusoidal_synth(amp, f0, 4096, 16000, harmonic=True))
    def step():
        amp.grad = None; f0.grad = None
        spectra.sinusoidal_synth(amp, f0, 4096, 16000, harmonic=True).backward(g)
    fb = timed(step)
    print(f"K = {K:2d}: forward {fwd:6.1f} us, forward + backward {fb:6.1f} us  ({fwd / K:5.2f} / {fb / K:5.2f} us per partial)")
