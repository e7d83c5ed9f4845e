"""oracle/make_golden_synth.py -- fixtures for the harmonic batch generator, FROM THE REFERENCE (build
container only; imports /root/reference behind the shim of oracle/make_golden.py).

  * dataset items: `synthetic_data.SimpleSinusoidDataset` (synthetic_data.py:76-118 parameters, :174-201 generate_sinusoids,
    :232-237 item normalisation) with the paper's settings (harmonic, 8 partials, f0 in [40, 1950] Hz, amplitudes in [0.4, 1],
    n_sinusoids_min 1, 4096 samples @ 16 kHz) after torch.manual_seed(SEED): frequency / weights of EVERY item of the first
    batch of 256 (the draws of the global torch RNG) and the audio `x` of the first 12;
  * the envelope upsamplers the synthesiser is built on (synths.py:95-113 -> ddsp.resample, ddsp.py:53-205): 'window'
    (overlapping Hann windows, add_endpoint) for amplitudes and 'bilinear' for frequencies, on time-varying random frames;
  * `synths.Sinusoidal(harmonic=True)` end to end on time-varying controls (two clips).
Writes tests/golden/synth_generator.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_synth.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.make_golden import import_reference, OUT  # noqa: E402

SEED = 4242


def main():
    import_reference()
    import ddsp  # type: ignore
    import synthetic_data  # type: ignore
    import synths  # type: ignore
    out = {"seed": np.int64(SEED)}
    torch.manual_seed(SEED)
    ds = synthetic_data.SimpleSinusoidDataset(freq_gen_min=40, freq_gen_max=1950, n_samples=4096, amplitude_min=0.4, amplitude_max=1,
                                              size=520, batch_size=8, batch_size_val=8, n_sinusoids=8, eval_split=0.2, test_split=0.1,
                                              n_sinusoids_min=1, harmonic=True)
    ds.setup()
    items = [ds.data[i] for i in range(256)]
    out["frequency"] = torch.stack([it["frequency"] for it in items]).numpy()      # [256, 1]
    out["weights"] = torch.stack([it["weights"] for it in items]).numpy()          # [256, 8] (masked amplitudes)
    out["x"] = torch.stack([items[i]["x"] for i in range(12)]).numpy()             # [12, 4096], peak 0.9
    print("items:", out["frequency"].shape, out["weights"].shape, out["x"].shape, "active partials of the first items:",
          (out["weights"][:8] > 0).sum(1))
    g = torch.Generator().manual_seed(SEED + 1)
    amp_frames = torch.rand(2, 16, 8, generator=g)
    freq_frames = 100 + 3000 * torch.rand(2, 16, 8, generator=g)
    out["amp_frames"], out["freq_frames"] = amp_frames.numpy(), freq_frames.numpy()
    out["amp_window_4096"] = ddsp.resample(amp_frames, 4096, method="window", add_endpoint=True).numpy()
    out["freq_bilinear_4096"] = ddsp.resample(freq_frames, 4096).numpy()
    f0_frames = 80 + 1500 * torch.rand(2, 16, 1, generator=g)
    synth = synths.Sinusoidal(4096, sample_rate=16000, amp_scale_fn=None, freq_scale_fn=None, harmonic=True)
    out["f0_frames"] = f0_frames.numpy()
    out["synth_audio"] = synth(amp_frames, f0_frames).numpy()
    path = os.path.join(OUT, "synth_generator.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
