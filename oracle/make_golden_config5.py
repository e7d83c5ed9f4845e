"""oracle/make_golden_config5.py -- BASELINE config 5 at its full size, FROM THE REFERENCE (build container only).

256 + 256 seeded harmonic clips (oracle/inputs.exact_harmonic_clips: bit-identical on every host, so only their sha256 is
stored) -> the reference's TorchSTFT (features.py:85-113; n_fft 2048, hop 256, flattop window) -> 4096 rows x 1025 bins ->
losses.Wasserstein1D in the paper's mode (SOT-2048 YAML: p=2, square_dist, dont_normalize, limit_quantile_range) with the
trainer's unit-scaled frequency positions (trainer.py:188-191) -> scalar, and its gradient with respect to the ESTIMATE's
audio through torch.stft's autograd (a strided sample: every 61st sample of every clip).
Also stores the 4096 row losses: a test can then tell the clips whose rows all sit away from the cutoff's knife edge
 from the ones where the two STFT implementations' last-bit differences flip a level.
Writes tests/golden/config5_256.npz (~85 KB).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_config5.py
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
from oracle.inputs import exact_harmonic_clips, sha256_of  # noqa: E402

SEED, CLIPS, STRIDE = 2026, 256, 61


def main():
    losses, features, _ = import_reference()
    target, estimate = exact_harmonic_clips(CLIPS, SEED)
    tfm = features.get_transform({"type": "stft", "n_fft": 2048, "hop_length": 256, "window": "flattop"}, 16000)
    est = estimate.clone().requires_grad_(True)
    sx, sy = tfm(target), tfm(est)
    pos = tfm.get_frequencies()
    pos = (pos / pos.max()).float()
    mod = losses.Wasserstein1D(**MODES["cutoff"])
    loss = mod(sx, sy, x_pos=pos, y_pos=pos.clone())
    (g_audio,) = torch.autograd.grad(loss, [est])
    # row losses of the same spectra (no mean): a per-row view for diagnosing a mismatch
    with torch.no_grad():
        rows = losses.Wasserstein1D(**MODES["cutoff"])(sx.reshape(-1, 1, 1025), sy.reshape(-1, 1, 1025), x_pos=pos, y_pos=pos.clone(),
                                                       dims=[1])
    out = dict(seed=np.int64(SEED), clips=np.int64(CLIPS), stride=np.int64(STRIDE),
               inputs_sha256=np.frombuffer(bytes.fromhex(sha256_of(target, estimate)), dtype=np.uint8),
               spec_shape=np.array(sx.shape, dtype=np.int64), loss=loss.detach().numpy(),
               grad_audio_sample=g_audio[:, ::STRIDE].contiguous().numpy(), grad_audio_peak=g_audio.abs().max().numpy(),
               grad_audio_l2=g_audio.double().norm().numpy(), row_loss=rows.reshape(-1).contiguous().numpy())
    path = os.path.join(OUT, "config5_256.npz")
    np.savez_compressed(path, **out)
    print("config 5:", tuple(sx.shape), "loss", float(loss), "grad peak", float(g_audio.abs().max()), "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
