"""Randomised sweep of the STFT-magnitude producer on the GPU box: HIP kernels (forward, backward from the stored spectrum and
recomputing, pair launch through MSSLoss-style use) against torch.stft + autograd on the same device, over n_fft, hop, clip length,
batch and window.
    python tools/fuzz_stft.py [seconds=60] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sot_amd import spectra
from sot_amd import _native as nat

nat.load(build_if_missing=False)
dev = torch.device("cuda:0")


def run(budget=60.0, seed0=0, max_cases=None, verbose=True):
    rng = np.random.default_rng(seed0)
    failures, cases, worst_f, worst_b = [], 0, 0.0, 0.0
    t_end = time.time() + budget
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        n_fft = int(2 ** rng.integers(6, 13))
        hop = int(rng.choice([n_fft // 8, n_fft // 4, n_fft // 2, n_fft, int(rng.integers(1, n_fft + 1))]))
        hop = max(1, hop)
        samples = int(rng.choice([1, hop, n_fft - 1, n_fft, n_fft + 1, int(rng.integers(1, 6 * n_fft + 2)), int(rng.integers(1, 20000))]))
        batch = int(rng.integers(1, 9))
        large = rng.random() < 0.15   # round 4: batches big enough for the one-wave-per-frame kernels of n_fft 2048 (forward >= 3072 frames,
        if large:                     # backward >= 1024 frame groups or >= 64 clips of <= 16 frames)
            n_fft = 2048
            hop = int(rng.choice([256, 512, 128, 300, 1024]))
            samples = int(rng.choice([4096, 4001, 3000, int(rng.integers(2048, 6000))]))
            batch = int(rng.integers(64, 321))
        if not spectra.hip_stft_supported(n_fft, hop, samples) or (not large and batch * (-(-samples // hop)) * n_fft > 6_000_000):
            continue
        window = str(rng.choice(["flattop", "hann", "blackman"]))
        seed = int(rng.integers(0, 2 ** 31 - 1))
        g = torch.Generator(device=dev).manual_seed(seed)
        kind = str(rng.choice(["noise", "tone", "sparse"]))
        audio = torch.randn(batch, samples, device=dev, generator=g)
        if kind == "tone":
            t = torch.arange(samples, device=dev) / 16000.0
            audio = torch.sin(2 * np.pi * 440.0 * t)[None].repeat(batch, 1) + 0.01 * audio
        elif kind == "sparse":
            audio = audio * (torch.rand(batch, samples, device=dev, generator=g) < 0.05)
        desc = dict(seed=seed, n_fft=n_fft, hop=hop, samples=samples, batch=batch, window=window, kind=kind, large=bool(large))
        a1 = audio.clone().requires_grad_(True)
        a2 = audio.clone().requires_grad_(True)
        a3 = audio.clone().requires_grad_(True)
        mag = spectra.stft_magnitude(a1, n_fft, hop, window)
        ref = spectra.stft_magnitude_torch(a2, n_fft, hop, window)
        if mag.shape != ref.shape:
            failures.append(("SHAPE", desc, tuple(mag.shape), tuple(ref.shape)))
            verbose and print("SHAPE", desc, tuple(mag.shape), tuple(ref.shape))
            cases += 1
            continue
        scale = float(ref.detach().abs().max()) + 1e-30
        ef = float((mag - ref).abs().max()) / scale
        wgt = torch.rand(ref.shape, device=dev, generator=g)
        (mag * wgt).sum().backward()
        (ref * wgt).sum().backward()
        spectra.SAVE_SPECTRUM = False
        try:
            (spectra.stft_magnitude(a3, n_fft, hop, window) * wgt).sum().backward()
        finally:
            spectra.SAVE_SPECTRUM = True
        gscale = float(a2.grad.abs().max()) + 1e-30
        eb = float((a1.grad - a2.grad).abs().max()) / gscale
        same = bool(torch.equal(a1.grad, a3.grad))
        frames_total = int(mag.shape[0] * mag.shape[1])
        if large or (n_fft == 2048 and frames_total >= 512):   # two factorisations of the transform at these sizes (round 5: the stored
            # spectrum of 512 ... 3071 frames comes from the wavefront-FFT kernel, the recomputing backward from the slot kernel): agreement to rounding, not bit for bit -- and like the agreement with
            # torch.stft's autograd (eb <= 2e-3) it is bounded by the bins with |X| ~ 0, whose X / |X| amplifies the last bits (pure tones:
            # up to ~1e-4 of the largest entry in the round-4 campaigns)
            same = float((a1.grad - a3.grad).abs().max()) <= 1e-3 * gscale
        if eb > 2e-3:   # a bin with |X| ~ 0 (pure tones: 1e-8 of the peak) makes X / |X| noise in float32 -- for torch.stft's autograd as well: the
            # float64 chain is the yardstick then, and the HIP gradient must be about as close to it as the reference's float32 gradient is
            a4 = audio.double().clone().requires_grad_(True)
            pad = spectra.end_padded(a4, n_fft, hop)
            spec64 = torch.stft(pad, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=spectra.analysis_window(window, n_fft, dev).double(),
                             center=False, normalized=True, return_complex=True).abs().permute(0, 2, 1)
            (spec64 * wgt.double()).sum().backward()
            err_hip = float((a1.grad.double() - a4.grad).abs().max()) / gscale
            err_ref = float((a2.grad.double() - a4.grad).abs().max()) / gscale
            if err_hip <= 4.0 * err_ref + 1e-5:
                eb = min(eb, 2e-3)
        worst_f, worst_b = max(worst_f, ef), max(worst_b, eb)
        cases += 1
        if not (ef <= 2e-5 and eb <= 2e-3 and same and bool(torch.isfinite(a1.grad).all())):
            failures.append(("STFT", desc, ef, eb, same))
            verbose and print("STFT", desc, "forward err / max", ef, "gradient err / max", eb, "stored-spectrum backward == recomputing backward:", same)
    if verbose:
        print(f"cases {cases}, outside tolerance {len(failures)}, worst forward err / max {worst_f:.3g}, worst gradient err / max {worst_b:.3g}")
    return cases, failures, worst_f, worst_b


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
