"""oracle/make_golden_dyadic.py -- full-size fixtures on DYADIC inputs, FROM THE REFERENCE (build container only).

SURVEY Appendix B.1 (iii): weights k/32 (and their squares k^2/1024) make every partial row sum exact in float32 whatever the
summation order, so the row mass S -- and with it the knife edge of the paper's cutoff mode -- is the same number for the
reference's torch.sum, for the HIP kernels and for the CSR form of the same rows.  Comparisons that are only "bulk" on random
inputs become row-for-row on these:

  * config 4 (8192 x 512, per-row amplitude cutoff -> ragged supports): the reference on the zero-masked dense rows, paper
    mode and p = 1: scalar + all 8192 row losses (masked-dense kernel AND the CSR kernel must match them row for row);
  * the SOT stage of config 5 (4096 rows x 1025 bins, rfftfreq positions), paper mode: scalar, the 4096 row losses and a
    strided sample of d loss / d y (every 41st bin).

Inputs are regenerated from the seed by sot_amd.bench_inputs.dyadic_* (sha256 stored).  Writes tests/golden/dyadic_full_size.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_dyadic.py
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
from oracle.inputs import sha256_of  # noqa: E402
from sot_amd.bench_inputs import dyadic_pairs, dyadic_ragged_supports  # noqa: E402

SEED, STRIDE = 4242, 41


def main():
    losses, _, _ = import_reference()
    out = dict(seed=np.int64(SEED), stride=np.int64(STRIDE))

    # ---- config 4, dyadic
    rs = dyadic_ragged_supports(8192, 512, SEED)
    xm, ym = rs["dense"]
    pos = rs["pos"]
    out["c4_inputs_sha256"] = np.frombuffer(bytes.fromhex(sha256_of(xm, ym)), dtype=np.uint8)
    for mode in ("cutoff", "p1"):
        mod = losses.Wasserstein1D(**MODES[mode])
        with torch.no_grad():
            rows = mod(xm.reshape(-1, 1, 512), ym.reshape(-1, 1, 512), x_pos=pos, y_pos=pos.clone(), dims=[1])
            scalar = mod(xm, ym, x_pos=pos, y_pos=pos.clone())
        out[f"c4_{mode}_rows"] = rows.reshape(-1).contiguous().numpy()
        out[f"c4_{mode}_scalar"] = scalar.numpy()
        print("config 4 dyadic", mode, float(scalar), "kept", rs["kept"])

    # ---- config 5's SOT stage, dyadic spectra
    x, y = dyadic_pairs(4096, 1025, SEED + 1)
    f = torch.fft.rfftfreq(2048, d=1.0 / 16000.0)
    p5 = (f / f.max()).float()
    out["c5_inputs_sha256"] = np.frombuffer(bytes.fromhex(sha256_of(x, y)), dtype=np.uint8)
    mod = losses.Wasserstein1D(**MODES["cutoff"])
    yv = y.clone().requires_grad_(True)
    loss = mod(x, yv, x_pos=p5, y_pos=p5.clone())
    (gy,) = torch.autograd.grad(loss, [yv])
    with torch.no_grad():
        rows = mod(x.reshape(-1, 1, 1025), y.reshape(-1, 1, 1025), x_pos=p5, y_pos=p5.clone(), dims=[1])
    out["c5_scalar"] = loss.detach().numpy()
    out["c5_rows"] = rows.reshape(-1).contiguous().numpy()
    out["c5_grad_y_sample"] = gy[:, ::STRIDE].contiguous().numpy()
    out["c5_grad_y_rowmax"] = gy.abs().amax(1).numpy()
    print("config 5 stage dyadic", float(loss))
    path = os.path.join(OUT, "dyadic_full_size.npz")
    np.savez_compressed(path, **out)
    print("->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
