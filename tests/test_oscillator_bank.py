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


def _cpu_envelopes(amp, freq, n_samples, sample_rate, harmonic):
    """The reference's composition on CPU torch ops (pinned bit for bit by test_generator_parameters_and_upsamplers_match_
    reference_on_cpu): harmonic frequencies, Nyquist mask, window / linear upsampling."""
    from sot_amd import spectra
    if harmonic:
        k = amp.shape[-1]
        freq = freq * torch.linspace(1.0, float(k), k, dtype=freq.dtype)
    amp = torch.where(freq >= sample_rate / 2.0, torch.zeros_like(amp), amp)
    if amp.dtype == torch.float64:   # the float64 form of the same two upsamplers, for gradient checks
        hop = n_samples // amp.shape[1]
        held = torch.cat([amp, amp[:, -1:, :]], dim=1)
        w = torch.hann_window(2 * hop).double()
        a = (held[:, :-1, None, :] * w[None, None, hop:, None] + held[:, 1:, None, :] * w[None, None, :hop, None]).reshape(amp.shape[0], n_samples, -1)
        f = torch.nn.functional.interpolate(freq.permute(0, 2, 1), size=n_samples, mode="linear", align_corners=False).permute(0, 2, 1)
        return a, f
    return spectra.upsample_window(amp, n_samples), spectra.upsample_linear(freq, n_samples)


def test_envelope_entry_points_validate_on_the_host():
    import ctypes
    from sot_amd import _native as nat
    lib = nat.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    call = lambda batch, frames, k, samples, rate=16000.0, a=p: lib.sot_synth_envelopes_forward(a, p, p, batch, frames, k, 0, samples, rate, p, p, None)
    assert call(1, 16, 2, 4100) == nat.SOT_ERR_BAD_SHAPE          # not whole hops (ddsp.py:163-170)
    assert call(1, 16, 2, 16) == nat.SOT_ERR_BAD_SHAPE            # not an upsampling (ddsp.py:155-160)
    assert call(1, 0, 2, 16) == nat.SOT_ERR_BAD_SHAPE and call(1, 4, 0, 16) == nat.SOT_ERR_BAD_SHAPE and call(1, 4, 2, 16, rate=0.0) == nat.SOT_ERR_BAD_SHAPE
    assert call(1, 4, 513, 16) == nat.SOT_ERR_UNSUPPORTED_SIZE and call(1, 4, 2, 1 << 21) == nat.SOT_ERR_UNSUPPORTED_SIZE
    assert call(0, 4, 2, 16) == nat.SOT_OK                        # an empty batch is a no-op
    assert call(1, 4, 2, 16, a=None) == nat.SOT_ERR_NULL_POINTER
    assert lib.sot_synth_envelopes_backward(p, p, p, 1, 4, 2, 0, 16, 16000.0, None, None, p, None, None) == nat.SOT_ERR_NULL_POINTER


@pytest.mark.gpu
def test_hip_envelopes_match_reference_bit_for_bit():
    """sot_synth_envelopes_forward against the reference's own ddsp.resample outputs (tests/golden/synth_generator.npz) and, with
    harmonic frequencies and partials above Nyquist, against the CPU composition pinned to the reference: identical bits."""
    from sot_amd import _native as nat
    fx = _synth_fixture()
    dev = torch.device("cuda:0")
    hann = torch.hann_window(512).to(dev)
    a, f = nat.synth_envelopes_forward(torch.tensor(fx["amp_frames"]).to(dev), torch.tensor(fx["freq_frames"]).to(dev), hann, 4096, 16000.0, False)
    assert np.array_equal(a.cpu().numpy(), fx["amp_window_4096"]) and np.array_equal(f.cpu().numpy(), fx["freq_bilinear_4096"])
    amp, f0 = torch.tensor(fx["amp_frames"]), torch.tensor(fx["f0_frames"])
    assert float((f0 * 8).max()) > 8000 > float(f0.min())           # some partials are above Nyquist in some frames only
    a, f = nat.synth_envelopes_forward(amp.to(dev), f0.to(dev), hann, 4096, 16000.0, True)
    wa, wf = _cpu_envelopes(amp, f0, 4096, 16000.0, True)
    assert torch.equal(a.cpu(), wa) and torch.equal(f.cpu(), wf)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,frames,k,samples,harmonic", [(1, 1, 1, 2, False), (3, 16, 8, 4096, True), (2, 7, 5, 7 * 33, False), (2, 250, 60, 1000, True),
                                                              (1, 4, 300, 64, False), (5, 64, 8, 64 * 250, True), (2, 3, 512, 12, True)])
def test_hip_envelopes_forward_and_backward_against_torch(batch, frames, k, samples, harmonic):
    from sot_amd import _native as nat, spectra
    g = torch.Generator().manual_seed(batch * 1000 + frames + k)
    amp = torch.rand(batch, frames, k, generator=g)
    freq = (60 + 2500 * torch.rand(batch, frames, 1, generator=g)) if harmonic else 100 + 9000 * torch.rand(batch, frames, k, generator=g)
    dev = torch.device("cuda:0")
    hann = torch.hann_window(2 * (samples // frames)).to(dev)
    a, f = nat.synth_envelopes_forward(amp.to(dev), freq.to(dev), hann, samples, 16000.0, harmonic)
    wa, wf = _cpu_envelopes(amp, freq, samples, 16000.0, harmonic)
    assert torch.equal(a.cpu(), wa) and torch.equal(f.cpu(), wf)
    # backward: against CPU autograd of the same composition -- in float32 (the reference's own arithmetic: the same
    # interpolation weights, sums in another order) and in float64 (whose weights differ from the float32 ones by ~frames * 6e-8)
    ga_env, gf_env = torch.randn(batch, samples, k, generator=g), torch.randn(batch, samples, k, generator=g)
    ga, gf = nat.synth_envelopes_backward(amp.to(dev), freq.to(dev), hann, samples, 16000.0, harmonic, ga_env.to(dev), gf_env.to(dev))
    for dtype, tol in ((torch.float32, 4e-6), (torch.float64, 2e-6 + 2e-7 * frames)):
        ampr, freqr = amp.detach().clone().to(dtype).requires_grad_(True), freq.detach().clone().to(dtype).requires_grad_(True)
        ea, ef = _cpu_envelopes(ampr, freqr, samples, 16000.0, harmonic)
        ((ea * ga_env.to(dtype)).sum() + (ef * gf_env.to(dtype)).sum()).backward()
        for got, want in ((ga, ampr.grad), (gf, freqr.grad)):
            assert got.shape == want.shape
            assert float((got.cpu().double() - want.double()).abs().max()) <= tol * max(1.0, float(want.abs().max())), (dtype, tol)
    # deterministic, and either gradient alone
    ga2, none = nat.synth_envelopes_backward(amp.to(dev), freq.to(dev), hann, samples, 16000.0, harmonic, ga_env.to(dev), None, need_freq=False)
    assert none is None and torch.equal(ga2, ga)
    none, gf2 = nat.synth_envelopes_backward(amp.to(dev), freq.to(dev), hann, samples, 16000.0, harmonic, None, gf_env.to(dev), need_amp=False)
    assert none is None and torch.equal(gf2, gf)
    # through the module function: autograd of the whole synthesiser vs the torch-op composition on the GPU
    if samples >= 64:
        a1, f1 = amp.to(dev).requires_grad_(True), freq.to(dev).requires_grad_(True)
        audio = spectra.sinusoidal_synth(a1, f1, samples, 16000, harmonic=harmonic)
        audio.square().mean().backward()
        a2, f2 = amp.to(dev).requires_grad_(True), freq.to(dev).requires_grad_(True)
        fr = f2 * torch.linspace(1.0, float(k), k, device=dev) if harmonic else f2
        am = torch.where(fr >= 8000.0, torch.zeros_like(a2), a2)
        want = spectra.oscillator_bank(spectra.upsample_linear(fr, samples), spectra.upsample_window(am, samples), 16000)
        want.square().mean().backward()
        assert float((audio - want).detach().abs().max()) <= 1e-4 * max(1.0, float(want.detach().abs().max()))
        assert float((a1.grad - a2.grad).abs().max()) <= 1e-4 * max(1e-6, float(a2.grad.abs().max()))
        assert float((f1.grad - f2.grad).abs().max()) <= 2e-3 * max(1e-9, float(f2.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("batch,frames,k,samples,harmonic", [(3, 16, 8, 4096, True), (2, 7, 5, 7 * 33, False), (2, 250, 60, 1000, True), (1, 4, 300, 64, False),
                                                              (5, 64, 8, 64 * 250, True), (1, 1, 1, 2, False), (256, 16, 8, 4096, True)])
def test_hip_synth_in_one_piece_equals_envelopes_then_bank(batch, frames, k, samples, harmonic):
    """sot_synth_forward / _backward (no sample-rate arrays) against the two-step HIP path: identical audio, bit for bit (same float32
    envelope values, same kernel after them); gradients to 2e-6 of their largest entry."""
    from sot_amd import _native as nat, spectra
    g = torch.Generator().manual_seed(batch * 77 + frames + k)
    dev = torch.device("cuda:0")
    amp = torch.rand(batch, frames, k, generator=g).to(dev)
    freq = ((60 + 2500 * torch.rand(batch, frames, 1, generator=g)) if harmonic else 100 + 9000 * torch.rand(batch, frames, k, generator=g)).to(dev)
    hann = torch.hann_window(2 * (samples // frames)).to(dev)
    a_env, f_env = nat.synth_envelopes_forward(amp, freq, hann, samples, 16000.0, harmonic)
    want = nat.oscillator_bank_forward(f_env, a_env, 16000.0)
    audio, ws = nat.synth_forward(amp, freq, hann, samples, 16000.0, harmonic, for_backward=True)
    assert torch.equal(audio, want)
    grad_audio = torch.randn(batch, samples, generator=g).to(dev)
    gf_env, ga_env = nat.oscillator_bank_backward(f_env, a_env, 16000.0, grad_audio)
    wa, wf = nat.synth_envelopes_backward(amp, freq, hann, samples, 16000.0, harmonic, ga_env, gf_env)
    def close(got, want, tol):   # the one-piece backward adds the same terms up in another order (fp64 accumulators either way)
        return float((got - want).abs().max()) <= tol * max(1e-30, float(want.abs().max()))
    first = None
    for reuse in (ws, None):
        ga, gf = nat.synth_backward(amp, freq, hann, samples, 16000.0, harmonic, grad_audio, forward_workspace=reuse)
        assert close(ga, wa, 2e-6) and close(gf, wf, 2e-6), (float((ga - wa).abs().max()), float((gf - wf).abs().max()), float(wf.abs().max()))
        first = first or (ga, gf)
        assert torch.equal(ga, first[0]) and torch.equal(gf, first[1])        # deterministic
    ga, none = nat.synth_backward(amp, freq, hann, samples, 16000.0, harmonic, grad_audio, need_freq=False)
    assert none is None and torch.equal(ga, first[0])
    none, gf = nat.synth_backward(amp, freq, hann, samples, 16000.0, harmonic, grad_audio, need_amp=False)
    assert none is None and torch.equal(gf, first[1])
    # the module function takes this path, with and without gradients
    assert spectra.FUSED_SYNTH
    a1, f1 = amp.clone().requires_grad_(True), freq.clone().requires_grad_(True)
    out = spectra.sinusoidal_synth(a1, f1, samples, 16000, harmonic=harmonic)
    assert torch.equal(out.detach(), want) and torch.equal(spectra.sinusoidal_synth(amp, freq, samples, 16000, harmonic=harmonic), want)
    (out * grad_audio).sum().backward()
    assert torch.equal(a1.grad, first[0]) and torch.equal(f1.grad, first[1])
