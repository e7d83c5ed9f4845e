"""oracle/make_golden_stft.py -- fixtures for the STFT-magnitude producer, generated FROM THE REFERENCE.

Runs only in the build container (imports /root/reference behind the shim of oracle/make_golden.py).  For seeded harmonic
clips it stores what the reference's `features.TorchSTFT` (features.py:85-113 -> compute_mag -> stft, :191-237;
utils.pad_for_stft, utils.py:252-275) returns, the scalar of the paper-cutoff SOT loss on those spectra, and the gradient
of that scalar with respect to the ESTIMATE's audio (through torch.stft's autograd): tests/golden/stft_chain.npz.
Also: MSSLoss (scalar and per-clip `dims` form) with its audio gradient, Wasserstein1DWithTransform, the oscillator bank, and the
paper's whole loss block (trainer.py:183-245: MixOfLosses of MSSLoss and Wasserstein1D) with its gradient.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_stft.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.make_golden import import_reference, MODES, OUT  # noqa: E402
from oracle.inputs import harmonic_audio_pair  # noqa: E402


def main():
    losses, features, _ = import_reference()
    out = {}
    # (n_fft, hop, samples): the paper's SOT-2048 and SOT-512 analysis settings, plus a clip whose length is not a
    # multiple of the hop (end padding) and is shorter than two frames
    for tag, n_fft, hop, n_samples, nb in (("a", 2048, 256, 4096, 2), ("b", 512, 128, 4096, 2), ("c", 1024, 256, 1000, 3)):
        ax, ay = harmonic_audio_pair(nb=nb, seed=100 + n_fft, n_samples=n_samples)
        tfm = features.get_transform({"type": "stft", "n_fft": n_fft, "hop_length": hop, "window": "flattop"}, 16000)
        ay_g = ay.clone().requires_grad_(True)
        sx, sy = tfm(ax), tfm(ay_g)
        pos = tfm.get_frequencies()
        pos = (pos / pos.max()).float()
        mod = losses.Wasserstein1D(**MODES["cutoff"])
        loss = mod(sx, sy, x_pos=pos, y_pos=pos.clone())
        (g_audio,) = torch.autograd.grad(loss, [ay_g])
        # plain sum of the magnitudes: a second, loss-independent pin of the STFT's own backward
        ay_h = ay.clone().requires_grad_(True)
        (g_sum,) = torch.autograd.grad(tfm(ay_h).sum(), [ay_h])
        out.update({f"{tag}_n_fft": np.int64(n_fft), f"{tag}_hop": np.int64(hop), f"{tag}_audio_x": ax.numpy(),
                    f"{tag}_audio_y": ay.numpy(), f"{tag}_spec_x": sx.detach().contiguous().numpy(),
                    f"{tag}_spec_y": sy.detach().contiguous().numpy(), f"{tag}_pos": pos.numpy(),
                    f"{tag}_loss": loss.detach().numpy(), f"{tag}_grad_audio_y": g_audio.numpy(),
                    f"{tag}_grad_sum_mag": g_sum.numpy()})
        print(tag, n_fft, hop, n_samples, tuple(sx.shape), float(loss))
    # MSSLoss (losses.py:365-425): scalar and gradient w.r.t. the estimate's audio, paper setting (mag only)
    # and with the log-magnitude term / L2
    ax, ay = harmonic_audio_pair(nb=2, seed=77, n_samples=4096)
    out["mss_audio_x"], out["mss_audio_y"] = ax.numpy(), ay.numpy()
    for tag, kw in (("paper", dict(mag_weight=1.0, logmag_weight=0.0)), ("both", dict(mag_weight=1.0, logmag_weight=0.5)),
                    ("l2", dict(mag_weight=0.7, logmag_weight=0.3, loss_type="L2"))):
        ay_g = ay.clone().requires_grad_(True)
        val = losses.MSSLoss(**kw)(ax, ay_g)
        (gr,) = torch.autograd.grad(val, [ay_g])
        out[f"mss_{tag}_loss"], out[f"mss_{tag}_grad_y"] = val.detach().numpy(), gr.numpy()
        print("MSSLoss", tag, float(val))
        # the same with `dims` = the two spectrogram axes (losses.py:406-425 hands it to mean_difference): one value per clip, and
        # the gradient of a weighted sum of them (round-4 review: the per-clip form was pinned to the package's own composition only)
        ay_c = ay.clone().requires_grad_(True)
        per_clip = losses.MSSLoss(**kw)(ax, ay_c, dims=(1, 2))
        wclip = torch.linspace(0.5, 1.5, ax.shape[0])
        (gc,) = torch.autograd.grad((per_clip * wclip).sum(), [ay_c])
        out[f"mss_{tag}_clip_loss"], out[f"mss_{tag}_clip_grad_y"], out["mss_clip_weights"] = per_clip.detach().numpy(), gc.numpy(), wclip.numpy()
    # the paper's loss block (trainer.py:183-245 with paper-experiments/SOT-2048/*/train_config.yaml:73-102): fresh unit-scaled bin
    # frequencies, TorchSTFT (n_fft 2048, hop 256, flattop) of both signals, MixOfLosses([MSSLoss(6 scales, L1, mag_weight 1),
    # Wasserstein1D(paper kwargs)], [0.05, 1]) -- MSSLoss on the audio, Wasserstein1D on the spectra -- total = sum of value.mean();
    # the trainer itself needs Lightning / wandb, so its lines are restated here around the reference's own modules
    ax, ay = harmonic_audio_pair(nb=4, seed=2048, n_samples=4096)
    tfm = features.get_transform({"type": "stft", "n_fft": 2048, "hop_length": 256, "window": "flattop"}, 16000)
    mix = losses.MixOfLosses([losses.MSSLoss(fft_sizes=(2048, 1024, 512, 256, 128, 64), loss_type="L1", mag_weight=1, logmag_weight=0),
                              losses.Wasserstein1D(**MODES["cutoff"], require_sort=True)], [0.05, 1])
    ay_g = ay.clone().requires_grad_(True)
    x_pos = torch.tensor(tfm.get_frequencies())
    x_pos = x_pos / x_pos.max()
    y_pos = x_pos.clone()
    spec_x, spec_x_hat = tfm(ax), tfm(ay_g)
    distance = {}
    for loss_fn, weight in zip(mix.losses, mix.weights):
        name = loss_fn.__class__.__name__
        a, b = (ax, ay_g) if name == "MSSLoss" else (spec_x, spec_x_hat)
        distance[name] = loss_fn(a, b, x_pos=x_pos, y_pos=y_pos) * weight
    total = 0
    for value in distance.values():
        total = total + value.mean()
    (gstep,) = torch.autograd.grad(total, [ay_g])
    out.update({"step_audio_x": ax.numpy(), "step_audio_y": ay.numpy(), "step_loss": total.detach().numpy(), "step_grad_y": gstep.numpy(),
                "step_mss_term": distance["MSSLoss"].detach().numpy(), "step_sot_term": distance["Wasserstein1D"].detach().numpy()})
    print("paper loss block:", float(total), {k: float(v) for k, v in distance.items()})
    # Wasserstein1DWithTransform (losses.py:316-343): audio in, TorchSTFT of both signals inside the module; a plain p = 1
    # configuration (hann window by default) and the paper's keyword set with a named window
    ax, ay = harmonic_audio_pair(nb=3, seed=31, n_samples=3000)
    out["wt_audio_x"], out["wt_audio_y"] = ax.numpy(), ay.numpy()
    for tag, p, tk, kw in (("p1", 1, {"type": "stft", "n_fft": 1024, "hop_length": 256, "sr": 16000}, {}),
                           ("paper", 2, {"type": "stft", "n_fft": 512, "hop_length": 128, "window": "flattop", "sr": 22050},
                            dict(square_dist=True, dont_normalize=True, limit_quantile_range=True))):
        ay_g = ay.clone().requires_grad_(True)
        val = losses.Wasserstein1DWithTransform(p=p, transform_kwargs=dict(tk), **kw)(ax, ay_g)
        (gr,) = torch.autograd.grad(val, [ay_g])
        out[f"wt_{tag}_loss"], out[f"wt_{tag}_grad_y"] = val.detach().numpy(), gr.numpy()
        print("Wasserstein1DWithTransform", tag, float(val))
    # oscillator bank (ddsp.py:208-263): audio and gradients w.r.t. both envelopes for seeded envelopes, one
    # sinusoid crossing Nyquist; T = 5000 is not a multiple of the kernel's time tile
    import ddsp  # type: ignore
    gen = torch.Generator().manual_seed(9)
    for tag, (nb, T, K) in (("o1", (2, 5000, 8)), ("o2", (3, 700, 3))):
        f0 = 60 + 900 * torch.rand(nb, 1, 1, generator=gen)
        glide = 1 + 0.2 * torch.linspace(0, 1, T).view(1, T, 1) * torch.rand(nb, 1, 1, generator=gen)
        freq = (f0 * glide * torch.arange(1, K + 1).view(1, 1, K)).float()
        freq[0, :, K - 1] = torch.linspace(7000, 9000, T)          # crosses sample_rate / 2
        amp = (torch.rand(nb, T, K, generator=gen) * torch.linspace(1, 0.2, K).view(1, 1, K)).float()
        fr, am = freq.clone().requires_grad_(True), amp.clone().requires_grad_(True)
        audio = ddsp.oscillator_bank(fr, am, sample_rate=16000)
        up = torch.randn(nb, T, generator=gen)
        gf, ga = torch.autograd.grad((audio * up).sum(), [fr, am])
        out.update({f"{tag}_freq": freq.numpy(), f"{tag}_amp": amp.numpy(), f"{tag}_audio": audio.detach().numpy(), f"{tag}_up": up.numpy(),
                    f"{tag}_grad_freq": gf.numpy(), f"{tag}_grad_amp": ga.numpy()})
        print("oscillator_bank", tag, tuple(audio.shape), float(audio.abs().max()))
    np.savez_compressed(os.path.join(OUT, "stft_chain.npz"), **out)
    print("wrote", os.path.join(OUT, "stft_chain.npz"), os.path.getsize(os.path.join(OUT, "stft_chain.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
