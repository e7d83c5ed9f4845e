"""oracle/inputs.py -- TEST INFRASTRUCTURE: seeded synthetic spectra shared by the golden
generator, the tests and bench.py (CPU generator => identical on every host with the same torch).
Distributions follow SURVEY.md §8(d): uniform U[0,1), "peaky" U^8 (close to real harmonic
spectra), dyadic k/32 (row sums exact in fp32 under ANY summation order), and edge rows."""
from __future__ import annotations

import hashlib

import numpy as np
import torch


def gen_inputs(kind, B, n, m, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, n, generator=g)
    y = torch.rand(B, m, generator=g)
    if kind == "uniform":
        pass
    elif kind == "peaky":
        x, y = x ** 8, y ** 8
    elif kind == "dyadic":
        g = torch.Generator().manual_seed(seed)
        x = torch.randint(0, 32, (B, n), generator=g).float() / 32
        y = torch.randint(0, 32, (B, m), generator=g).float() / 32
    elif kind == "edge":  # zero rows, Diracs, zero-weight runs, identical rows, sub-eps mass
        x, y = x ** 4, y ** 4
        x[0] = 0.0                                 # all-zero x row
        if B > 1:
            y[1] = 0.0                             # all-zero y row
        if B > 2:
            x[2] = 0.0
            x[2, n // 3] = 1.0                     # Dirac vs Dirac
            y[2] = 0.0
            y[2, (2 * m) // 3] = 2.5
        if B > 3:
            y[3, : m // 2] = 0.0                   # leading zero-weight run
            x[3, n // 2:] = 0.0                    # trailing zero-weight run
        if B > 4 and n == m:
            y[4] = x[4]                            # identical distributions
        if B > 5:
            x[5] = x[5] * 1e-9                     # mass below the 1e-7 guard of safe_divide
    else:
        raise ValueError(kind)
    return x.contiguous(), y.contiguous()


def positions(spec, n):
    """Position grids named in the manifest."""
    if spec == "linspace":
        return torch.linspace(0, 1, n)
    if spec.startswith("rfftfreq"):
        nfft = 2 * (n - 1)
        pos = torch.fft.rfftfreq(nfft, 1 / 16000.0)
        return (pos / pos.max()).float()
    raise ValueError(spec)


def sha256_of(*tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()


def harmonic_audio_pair(nb=2, seed=11, n_samples=4096, sr=16000.0):
    """Two batches of additive harmonic clips (f0 ~ U[40,1950] Hz, 8 partials, amps ~ U[0.4,1], partials above
    Nyquist muted, peak 0.9): the signal distribution of the reference's synthetic_data.py:331-345.  This exact
    code produced the audio behind tests/golden/inputs_harmonic_stft.npz (oracle/make_golden.py, section 8)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n_samples) / sr
    k = torch.arange(1, 9).view(1, 8, 1)

    def additive():
        f0 = 40 + (1950 - 40) * torch.rand(nb, 1, 1, generator=g)
        amps = 0.4 + 0.6 * torch.rand(nb, 8, 1, generator=g)
        audible = (f0 * k < sr / 2).float()
        sig = (amps * audible * torch.sin(2 * torch.pi * f0 * k * t.view(1, 1, -1))).sum(1)
        return 0.9 * sig / sig.abs().amax(dim=1, keepdim=True)

    return additive(), additive()


def exact_harmonic_clips(clips=256, seed=2026, n_samples=4096, sr=16000.0):
    """Harmonic clips that are BIT-IDENTICAL on every host: the parameter distribution of the reference's
    SimpleSinusoidDataset (synthetic_data.py:76-118: f0 ~ U[40,1950] Hz, 8 amplitudes ~ U[0.4,1], the partials after
    max(1, n_active), n_active ~ U{0..7}, muted; items peak-normalised to 0.9, :232-237), drawn from numpy's MT19937 and
    synthesised with IEEE basic operations only (+, -, *, floor, abs on float64; the sine is the parabola
    y = 4x(1-|x|), y += 0.225 (y|y| - y) of the phase wrapped to [-1, 1)), so no libm / SIMD variant enters.  Used where a
    fixture holds reference OUTPUTS for inputs too large to store (tests/golden/config5_256.npz stores their sha256).
    Returns (target, estimate) float32 [clips, n_samples]."""
    rs = np.random.RandomState(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    k = np.arange(1, 9, dtype=np.float64).reshape(1, 8, 1)

    def batch():
        f0 = 40.0 + (1950.0 - 40.0) * rs.rand(clips, 1, 1)
        amps = 0.4 + 0.6 * rs.rand(clips, 8, 1)
        n_active = rs.randint(0, 8, size=(clips, 1, 1))
        keep = ((k <= np.maximum(n_active, 1)) & (f0 * k < sr / 2)).astype(np.float64)
        cycles = f0 * k * t.reshape(1, 1, -1)
        x = 2.0 * (cycles - np.floor(cycles)) - 1.0          # phase in [-1, 1): sin(pi x) ~ parabola below
        y = 4.0 * x * (1.0 - np.abs(x))
        y = y + 0.225 * (y * np.abs(y) - y)
        sig = (amps * keep * y).sum(axis=1)
        peak = np.abs(sig).max(axis=1, keepdims=True)
        return torch.from_numpy((0.9 * sig / (peak + 1e-7)).astype(np.float32))

    return batch(), batch()
