"""Oscillator bank in front of the STFT: the HIP kernels and the CPU composition against outputs and
autograd gradients of the reference's ddsp.oscillator_bank (fixtures from oracle/make_golden_stft.py, keys o1_*, o2_*)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from sot_amd import spectra

TAGS = ("o1", "o2")


def _fx():
    return dict(np.load(os.path.join(GOLDEN, "stft_chain.npz")))


def _rel(got, want):
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


@pytest.mark.parametrize("tag", TAGS)
def test_cpu_composition_matches_reference(tag):
    fx = _fx()
    f = torch.from_numpy(fx[f"{tag}_freq"]).requires_grad_(True)
    a = torch.from_numpy(fx[f"{tag}_amp"]).requires_grad_(True)
    audio = spectra.oscillator_bank(f, a, 16000)
    assert _rel(audio.detach().numpy(), fx[f"{tag}_audio"]) <= 1e-6
    (audio * torch.from_numpy(fx[f"{tag}_up"])).sum().backward()
    assert _rel(f.grad.numpy(), fx[f"{tag}_grad_freq"]) <= 1e-5
    assert _rel(a.grad.numpy(), fx[f"{tag}_grad_amp"]) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_oscillator_bank_matches_reference(tag):
    fx = _fx()
    dev = torch.device("cuda:0")
    f = torch.from_numpy(fx[f"{tag}_freq"]).to(dev).requires_grad_(True)
    a = torch.from_numpy(fx[f"{tag}_amp"]).to(dev).requires_grad_(True)
    audio = spectra.oscillator_bank(f, a, 16000)
    # phases reach a few thousand radians, where one fp32 ulp is ~2.4e-4: sin() of the SAME fp32 phase is what is compared,
    # so the only differences are sinf's last ulp and the summation order over the sinusoids
    assert _rel(audio.detach().cpu().numpy(), fx[f"{tag}_audio"]) <= 2e-6
    (audio * torch.from_numpy(fx[f"{tag}_up"]).to(dev)).sum().backward()
    assert _rel(a.grad.cpu().numpy(), fx[f"{tag}_grad_amp"]) <= 2e-6
    assert _rel(f.grad.cpu().numpy(), fx[f"{tag}_grad_freq"]) <= 2e-5
    # sinusoids above Nyquist are muted: no gradient reaches their amplitude
    muted = fx[f"{tag}_freq"] >= 8000.0
    assert muted.any() and np.all(a.grad.cpu().numpy()[muted] == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,samples,k", [(1, 1, 1), (3, 2048, 4), (2, 2049, 7), (5, 4096, 60), (2, 20000, 17), (64, 4096, 8), (2, 700, 300), (1, 50, 512)])
def test_hip_oscillator_bank_against_torch_autograd(batch, samples, k):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(samples + k)
    f = (30 + 9000 * torch.rand(batch, samples, k, generator=g)).to(dev)
    a = torch.rand(batch, samples, k, generator=g).to(dev)
    up = torch.randn(batch, samples, generator=g).to(dev)
    f1, a1 = f.clone().requires_grad_(True), a.clone().requires_grad_(True)
    got = spectra.oscillator_bank(f1, a1, 16000)
    (got * up).sum().backward(retain_graph=True)
    first = (f1.grad.clone(), a1.grad.clone())
    f1.grad = a1.grad = None
    (got * up).sum().backward()   # the backward reuses the forward's segment phases: a second walk gives the same bits
    assert torch.equal(f1.grad, first[0]) and torch.equal(a1.grad, first[1])
    # the same composition in float64 on the fp32-rounded phases is not expressible; compare with the CPU composition
    f2, a2 = f.cpu().clone().requires_grad_(True), a.cpu().clone().requires_grad_(True)
    want = spectra.oscillator_bank(f2, a2, 16000)
    (want * up.cpu()).sum().backward()
    assert _rel(got.detach().cpu().numpy(), want.detach().numpy()) <= 3e-6
    assert _rel(a1.grad.cpu().numpy(), a2.grad.numpy()) <= 3e-6
    assert _rel(f1.grad.cpu().numpy(), f2.grad.numpy()) <= 3e-5
    # deterministic: a second evaluation is bit-identical
    again = spectra.oscillator_bank(f, a, 16000)
    assert torch.equal(again, got.detach())


@pytest.mark.gpu
def test_hip_oscillator_bank_only_amplitude_gradient_and_errors():
    from sot_amd import _native as nat
    dev = torch.device("cuda:0")
    f = torch.full((2, 300, 3), 440.0, device=dev)
    a = torch.rand(2, 300, 3, device=dev).requires_grad_(True)
    spectra.oscillator_bank(f, a, 16000).sum().backward()
    assert a.grad is not None and torch.isfinite(a.grad).all()
    with pytest.raises(RuntimeError):
        nat.oscillator_bank_forward(f, a.detach()[:, :, :2], 16000.0)
    with pytest.raises(RuntimeError):
        nat.oscillator_bank_forward(f.cpu(), a.detach().cpu(), 16000.0)


def _synth_fixture():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "synth_generator.npz"))


def test_generator_parameters_and_upsamplers_match_reference_on_cpu():
    """SURVEY 8f row 2, generator half, on the CPU (torch ops; the same functions run on the GPU): the parameter draws of
    SimpleSinusoidDataset.setup() (synthetic_data.py:76-118), the two envelope upsamplers of synths.py:95-113 and the
    whole item pipeline (generate_sinusoids :174-201 + item normalisation :232-237) against what the reference itself
    produced (oracle/make_golden_synth.py).  Bit for bit: same ATen kernels, same operation order."""
    from sot_amd import spectra
    fx = _synth_fixture()
    f, w = spectra.harmonic_parameters(520, int(fx["seed"]))
    assert np.array_equal(f[:256].numpy(), fx["frequency"]) and np.array_equal(w[:256].numpy(), fx["weights"])
    assert set((fx["weights"] > 0).sum(1).tolist()) <= set(range(1, 8))      # 1..7 sounding partials, never the 8th
    assert np.array_equal(spectra.upsample_window(torch.tensor(fx["amp_frames"]), 4096).numpy(), fx["amp_window_4096"])
    assert np.array_equal(spectra.upsample_linear(torch.tensor(fx["freq_frames"]), 4096).numpy(), fx["freq_bilinear_4096"])
    x = spectra.harmonic_items(torch.tensor(fx["frequency"][:12]), torch.tensor(fx["weights"][:12])).numpy()
    assert np.array_equal(x, fx["x"]) and abs(np.abs(x).max() - 0.9) < 1e-6
    audio = spectra.sinusoidal_synth(torch.tensor(fx["amp_frames"]), torch.tensor(fx["f0_frames"]), 4096).numpy()
    assert np.array_equal(audio, fx["synth_audio"])
    with pytest.raises(ValueError):
        spectra.upsample_window(torch.zeros(1, 16, 2), 4100)              # not a multiple of the frame count (ddsp.py:163-170)


@pytest.mark.gpu
def test_generator_on_the_gpu_matches_reference_items():
    """The same pipeline on the GPU: upsamplers on torch's device ops, oscillator bank on the HIP kernels.  Items and the
    time-varying synthesiser output against the reference's audio: <= 2e-5 of the peak (the phase of a 4096-sample clip reaches
    ~5e3 rad, so one ulp of a frequency envelope is worth ~3e-4 rad at its end)."""
    from sot_amd import spectra
    fx = _synth_fixture()
    dev = torch.device("cuda:0")
    a = spectra.upsample_window(torch.tensor(fx["amp_frames"]).to(dev), 4096).cpu().numpy()
    np.testing.assert_allclose(a, fx["amp_window_4096"], rtol=0, atol=3e-7)     # the device's hann window differs from the CPU's by an ulp
    b = spectra.upsample_linear(torch.tensor(fx["freq_frames"]).to(dev), 4096).cpu().numpy()
    np.testing.assert_allclose(b, fx["freq_bilinear_4096"], rtol=3e-7, atol=0)
    x = spectra.harmonic_items(torch.tensor(fx["frequency"][:12]).to(dev), torch.tensor(fx["weights"][:12]).to(dev)).cpu().numpy()
    assert np.abs(x - fx["x"]).max() <= 2e-5 * 0.9, float(np.abs(x - fx["x"]).max())
    audio = spectra.sinusoidal_synth(torch.tensor(fx["amp_frames"]).to(dev), torch.tensor(fx["f0_frames"]).to(dev), 4096).cpu().numpy()
    assert np.abs(audio - fx["synth_audio"]).max() <= 2e-5 * np.abs(fx["synth_audio"]).max()
    # the seeded batch generator = these parameters through that pipeline; gradients flow to the frame-rate controls
    clips = spectra.harmonic_batch(12, device=dev, seed=int(fx["seed"]))
    # (harmonic_parameters(12, seed) draws differ from the first 12 of a 520-item draw: only shapes / peak are checked here)
    assert clips.shape == (12, 4096) and abs(float(clips.abs().max()) - 0.9) < 1e-5
    amp = torch.tensor(fx["amp_frames"]).to(dev).requires_grad_(True)
    f0 = torch.tensor(fx["f0_frames"]).to(dev).requires_grad_(True)
    spectra.sinusoidal_synth(amp, f0, 4096).square().mean().backward()
    assert torch.isfinite(amp.grad).all() and torch.isfinite(f0.grad).all() and float(amp.grad.abs().max()) > 0
