"""oracle/make_golden_units.py -- fixture for the log-frequency position map of the reference's trainer (trainer.py:187-191:
`x_pos = hz_to_unit(x_pos, freq_hz_min, freq_hz_max)` when the loss carries `log_scaled_x`), generated FROM THE REFERENCE.

TEST INFRASTRUCTURE: runs only in the build container (imports /root/reference behind the shim of oracle/make_golden.py) and stores
what `utils.hz_to_unit` (utils.py:85-114) returns for the bin frequencies of two STFT settings and a CQT-like geometric grid:
tests/golden/hz_to_unit.npz (arrays only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_units.py
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


def main():
    import_reference()
    import utils  # type: ignore  (the reference's utils.py: /root/reference is on sys.path now)
    out = {}
    grids = {"stft2048": torch.fft.rfftfreq(2048, d=1.0 / 16000.0), "stft512": torch.fft.rfftfreq(512, d=1.0 / 16000.0),
             "geometric": 32.7 * 2.0 ** (torch.arange(96, dtype=torch.float32) / 12.0)}
    for tag, hz in grids.items():
        for k, (lo, hi, clip) in enumerate(((32.7, 8000.0, False), (20.0, 7000.0, True), (float(hz[1]), float(hz[-1]), False))):
            out[f"{tag}_{k}_hz"], out[f"{tag}_{k}_lo"], out[f"{tag}_{k}_hi"] = hz.numpy(), np.float64(lo), np.float64(hi)
            out[f"{tag}_{k}_clip"] = np.int64(clip)
            out[f"{tag}_{k}_unit"] = utils.hz_to_unit(hz, lo, hi, clip=clip).numpy()
    np.savez(os.path.join(OUT, "hz_to_unit.npz"), **out)
    print("wrote", os.path.join(OUT, "hz_to_unit.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
