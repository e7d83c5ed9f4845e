"""Oscillator bank (sot_oscillator_bank_*) against the torch composition of the same ops on the GPU.
usage: python tools/bench_osc.py [batch samples sinusoids]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sot_amd import spectra


def torch_bank(f, a, sr):
    a = torch.where(f >= sr / 2.0, torch.zeros_like(a), a)
    return torch.sum(a * torch.sin(torch.cumsum(f * (2.0 * torch.pi) / float(sr), dim=1)), dim=-1)


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    shapes = [(64, 4096, 8), (64, 4096, 64), (16, 64000, 100), (256, 4096, 16)]
    if len(sys.argv) == 4:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    dev = torch.device("cuda:0")
    for batch, samples, k in shapes:
        f = (40 + 7000 * torch.rand(batch, samples, k, device=dev)).requires_grad_(True)
        a = torch.rand(batch, samples, k, device=dev).requires_grad_(True)
        up = torch.randn(batch, samples, device=dev)

        def step(fn):
            f.grad = a.grad = None
            (fn(f, a, 16000) * up).sum().backward()

        with torch.no_grad():
            fwd_hip = timed(lambda: spectra.oscillator_bank(f, a, 16000))
            fwd_torch = timed(lambda: torch_bank(f, a, 16000))
        both_hip = timed(lambda: step(spectra.oscillator_bank))
        both_torch = timed(lambda: step(torch_bank))
        mb = batch * samples * k * 4 / 1e6
        print(f"[{batch}x{samples}x{k}] ({mb:.0f} MB/envelope) forward hip {fwd_hip:.3f} ms  torch {fwd_torch:.3f} ms | "
              f"fwd+bwd hip {both_hip:.3f} ms  torch {both_torch:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
