"""oracle/make_golden_stft.py -- fixtures for the STFT-magnitude producer, generated FROM THE REFERENCE.

Runs only in the build container (imports /root/reference behind the shim of oracle/make_golden.py).  For seeded harmonic
clips it stores what the reference's `features.TorchSTFT` (features.py:85-113 -> compute_mag -> stft, :191-237;
utils.pad_for_stft, utils.py:252-275) returns, the scalar of the paper-cutoff SOT loss on those spectra, and the gradient
of that scalar with respect to the ESTIMATE's audio (through torch.stft's autograd): tests/golden/stft_chain.npz.

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
