"""tools/synth_probe.py -- time the harmonic synthesiser (frame-rate controls -> audio) and its backward on the GPU:
envelope kernels + oscillator bank, next to the torch-op envelopes the module used before.
Usage: python tools/synth_probe.py [batch] [frames] [samples]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sot_amd import spectra  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    batch, frames, samples = (int(v) for v in (sys.argv[1:4] + [256, 16, 4096][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    amp = torch.rand(batch, frames, 8, device=dev)
    f0 = 40 + 1900 * torch.rand(batch, frames, 1, device=dev)

    def torch_envelopes(a, f):
        fr = f * torch.linspace(1.0, 8.0, 8, device=dev)
        am = torch.where(fr >= 8000.0, torch.zeros_like(a), a)
        return spectra.oscillator_bank(spectra.upsample_linear(fr, samples), spectra.upsample_window(am, samples), 16000)

    print(f"batch {batch} frames {frames} samples {samples}")
    print("forward, one piece        : %8.1f us" % timed(lambda: spectra.sinusoidal_synth(amp, f0, samples)))
    spectra.FUSED_SYNTH = False
    print("forward, envelope kernels : %8.1f us" % timed(lambda: spectra.sinusoidal_synth(amp, f0, samples)))
    spectra.FUSED_SYNTH = True
    print("forward, torch envelopes  : %8.1f us" % timed(lambda: torch_envelopes(amp, f0)))
    ar, fr = amp.clone().requires_grad_(True), f0.clone().requires_grad_(True)

    def step(fn):
        ar.grad = fr.grad = None
        fn(ar, fr).square().mean().backward()
    print("fwd+bwd, one piece        : %8.1f us" % timed(lambda: step(lambda a, f: spectra.sinusoidal_synth(a, f, samples))))
    spectra.FUSED_SYNTH = False
    print("fwd+bwd, envelope kernels : %8.1f us" % timed(lambda: step(lambda a, f: spectra.sinusoidal_synth(a, f, samples))))
    spectra.FUSED_SYNTH = True
    print("fwd+bwd, torch envelopes  : %8.1f us" % timed(lambda: step(torch_envelopes)))
    from sot_amd import _native as nat
    hann = torch.hann_window(2 * samples // frames).to(dev)
    print("envelope forward kernel   : %8.1f us" % timed(lambda: nat.synth_envelopes_forward(amp, f0, hann, samples, 16000.0, True)))
    g = torch.randn(batch, samples, 8, device=dev)
    print("envelope backward kernel  : %8.1f us" % timed(lambda: nat.synth_envelopes_backward(amp, f0, hann, samples, 16000.0, True, g, g)))
    print("synth forward (C ABI)     : %8.1f us" % timed(lambda: nat.synth_forward(amp, f0, hann, samples, 16000.0, True)))
    ga = torch.randn(batch, samples, device=dev)
    print("synth backward (C ABI)    : %8.1f us" % timed(lambda: nat.synth_backward(amp, f0, hann, samples, 16000.0, True, ga)))
    fe, ae = nat.synth_envelopes_forward(amp, f0, hann, samples, 16000.0, True)
    print("oscillator bank forward   : %8.1f us" % timed(lambda: nat.oscillator_bank_forward(fe, ae, 16000.0)))


if __name__ == "__main__":
    main()
