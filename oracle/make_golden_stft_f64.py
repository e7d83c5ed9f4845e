"""oracle/make_golden_stft_f64.py -- float64 evaluations of the chains pinned in tests/golden/stft_chain.npz, FROM THE REFERENCE
run in double precision (build container only; imports /root/reference behind the shim of oracle/make_golden.py).

The float32 fixtures of make_golden_stft.py carry the reference's own rounding: gradients through |STFT| are ill-conditioned in
noise-floor bins, so two correct float32 implementations differ there by ~1e-3 of the gradient's peak.  These float64 values are
the yardstick for that statement: tests compare |HIP - float64| with |reference float32 - float64| (tests/test_stft_producer.py).
Same clips, same modules; `torch.set_default_dtype(torch.float64)` so that the reference's windows are built in double, and the
reference's `torch_float32` casts (features.py:193,203,237) are replaced IN THIS PROCESS by casts to float64 -- the one deviation from
running the reference as it is, made here and nowhere else.
Writes tests/golden/stft_chain_f64.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_stft_f64.py
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
    clips = {}
    for tag, n_fft, hop, n_samples, nb in (("a", 2048, 256, 4096, 2), ("b", 512, 128, 4096, 2), ("c", 1024, 256, 1000, 3)):
        clips[tag] = (n_fft, hop) + harmonic_audio_pair(nb=nb, seed=100 + n_fft, n_samples=n_samples)   # float32 clips, as in the fixture
    mss = harmonic_audio_pair(nb=2, seed=77, n_samples=4096)
    wt = harmonic_audio_pair(nb=3, seed=31, n_samples=3000)
    torch.set_default_dtype(torch.float64)
    to_double = lambda t: t.type(torch.float64) if isinstance(t, torch.Tensor) else torch.tensor(t, dtype=torch.float64)  # noqa: E731
    features.torch_float32 = to_double
    out = {}
    for tag, (n_fft, hop, ax, ay) in clips.items():
        ax, ay = ax.double(), ay.double()
        tfm = features.get_transform({"type": "stft", "n_fft": n_fft, "hop_length": hop, "window": "flattop"}, 16000).double()
        ay_g = ay.clone().requires_grad_(True)
        sx, sy = tfm(ax), tfm(ay_g)
        assert sy.dtype == torch.float64
        pos = tfm.get_frequencies()
        pos = (pos / pos.max()).float().double()     # the float32 positions of the fixture, exactly
        loss = losses.Wasserstein1D(**MODES["cutoff"])(sx, sy, x_pos=pos, y_pos=pos.clone())
        (g_audio,) = torch.autograd.grad(loss, [ay_g])
        ay_h = ay.clone().requires_grad_(True)
        (g_sum,) = torch.autograd.grad(tfm(ay_h).sum(), [ay_h])
        out.update({f"{tag}_spec_y": sy.detach().contiguous().numpy(), f"{tag}_loss": loss.detach().numpy(),
                    f"{tag}_grad_audio_y": g_audio.numpy(), f"{tag}_grad_sum_mag": g_sum.numpy()})
        print(tag, float(loss), g_audio.dtype)
    ax, ay = mss[0].double(), mss[1].double()
    for tag, kw in (("paper", dict(mag_weight=1.0, logmag_weight=0.0)), ("both", dict(mag_weight=1.0, logmag_weight=0.5)),
                    ("l2", dict(mag_weight=0.7, logmag_weight=0.3, loss_type="L2"))):
        ay_g = ay.clone().requires_grad_(True)
        val = losses.MSSLoss(**kw).double()(ax, ay_g)
        (gr,) = torch.autograd.grad(val, [ay_g])
        out[f"mss_{tag}_loss"], out[f"mss_{tag}_grad_y"] = val.detach().numpy(), gr.numpy()
        print("MSSLoss", tag, float(val), gr.dtype)
    ax, ay = wt[0].double(), wt[1].double()
    for tag, p, tk, kw in (("p1", 1, {"type": "stft", "n_fft": 1024, "hop_length": 256, "sr": 16000}, {}),
                           ("paper", 2, {"type": "stft", "n_fft": 512, "hop_length": 128, "window": "flattop", "sr": 22050},
                            dict(square_dist=True, dont_normalize=True, limit_quantile_range=True))):
        ay_g = ay.clone().requires_grad_(True)
        val = losses.Wasserstein1DWithTransform(p=p, transform_kwargs=dict(tk), **kw).double()(ax, ay_g)
        (gr,) = torch.autograd.grad(val, [ay_g])
        out[f"wt_{tag}_loss"], out[f"wt_{tag}_grad_y"] = val.detach().numpy(), gr.numpy()
        print("Wasserstein1DWithTransform", tag, float(val), gr.dtype)
    path = os.path.join(OUT, "stft_chain_f64.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")
    # how far the reference's own float32 evaluation is from these values
    fx = np.load(os.path.join(OUT, "stft_chain.npz"))
    for k in sorted(out):
        if "grad" in k or "loss" in k:
            a, b = fx[k].astype(np.float64), out[k]
            print(f"  {k:22s} reference float32 vs float64: max {np.abs(a - b).max() / np.abs(b).max():.2e} of the peak, "
                  f"median {np.median(np.abs(a - b)) / np.abs(b).max():.2e}")


if __name__ == "__main__":
    main()
