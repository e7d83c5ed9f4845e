"""Seeded synthetic inputs of the BASELINE.json workloads (bench.py's timed inputs and the shapes its `extras` report).

CPU generator => identical tensors on every host with the same torch build.  `spectrum_pairs` is the recipe of
BASELINE.md §4 (`torch.Generator().manual_seed(seed)`, x drawn before y; "peaky" = U^8, close to real harmonic spectra);
`ragged_supports` is BASELINE config 4 (per-row amplitude cutoff tau_r = 10^U[-3,-0.3] * max_r -> variable supports) in
both input forms (zero-masked dense rows and CSR).  tests/test_host_api.py checks that `spectrum_pairs` draws exactly the
tensors the golden fixtures were computed on, so the stored reference scalars apply to them.
"""
from __future__ import annotations

import torch


def spectrum_pairs(kind: str, rows: int, n: int, m: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(rows, n, generator=g)
    y = torch.rand(rows, m, generator=g)
    if kind == "peaky":
        x, y = x ** 8, y ** 8
    elif kind != "uniform":
        raise ValueError(kind)
    return x.contiguous(), y.contiguous()


def dyadic_pairs(rows: int, n: int, seed: int):
    """Weights k / 32, k in 0..31 (SURVEY Appendix B.1 iii): they and their squares sum exactly in float32 in ANY order for rows of
    up to 2^14 points, so the row mass -- and the cutoff's knife edge -- does not depend on who sums.  x drawn before y."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 32, (rows, n), generator=g).float() / 32
    y = torch.randint(0, 32, (rows, n), generator=g).float() / 32
    return x.contiguous(), y.contiguous()


def dyadic_ragged_supports(rows: int = 8192, n: int = 512, seed: int = 4242):
    """BASELINE config 4's ragged supports on dyadic weights: the same dict as ragged_supports()."""
    x, y = dyadic_pairs(rows, n, seed)
    return _ragged(x, y, rows, n, seed)


def ragged_supports(rows: int = 8192, n: int = 512, seed: int = 1234):
    """BASELINE config 4.  Returns dict(dense=(xm, ym), csr=((xw, xp, xoff), (yw, yp, yoff)), max_n, max_m, pos, kept)
    on the CPU: masked-dense rows (entries below the row's threshold set to 0) and the same supports in CSR form
    (concatenated kept weights and their positions, int64 offsets [rows + 1])."""
    x, y = spectrum_pairs("peaky", rows, n, n, seed)
    return _ragged(x, y, rows, n, seed)


def _ragged(x, y, rows, n, seed):
    g = torch.Generator().manual_seed(seed)
    tau = 10 ** (-3 + 2.7 * torch.rand(rows, 1, generator=g))
    keep_x, keep_y = x >= tau * x.amax(1, keepdim=True), y >= tau * y.amax(1, keepdim=True)
    pos = torch.linspace(0, 1, n)

    def csr(dense, keep):
        off = torch.zeros(rows + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(keep.sum(1), 0)
        return dense[keep].contiguous(), pos.expand_as(dense)[keep].contiguous(), off

    zero = torch.zeros(())
    return {"dense": (torch.where(keep_x, x, zero).contiguous(), torch.where(keep_y, y, zero).contiguous()),
            "csr": (csr(x, keep_x), csr(y, keep_y)), "max_n": int(keep_x.sum(1).max()), "max_m": int(keep_y.sum(1).max()),
            "pos": pos, "kept": float(keep_x.sum() + keep_y.sum()) / (2 * rows)}
