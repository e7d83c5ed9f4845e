"""The loss on torch ops, for the tensors the HIP library does not take: CPU tensors (`accelerator: cpu`, Lightning's sanity
runs, unit tests of a training script on a laptop) and float64 tensors on any device.

The reference is device- and dtype-agnostic (losses.py:129-313 are plain ATen calls; utils.py:135-142 promotes), so a drop-in
has to be as well.  This module is that part of the drop-in: the same quantities in the same order of floating-point
operations as the reference -- hence the same bits on the same device type -- written as one pass over the merged quantile
levels.  It is NOT a fallback for the GPU float32 path: float32 tensors on a HIP device never come here (they fail loudly when
libsot_hip.so is missing), and nothing in this file comes from the test infrastructure.

Differentiable through torch autograd (weights and positions), like the reference.
"""
from __future__ import annotations

import torch


def guarded_ratio(values: torch.Tensor, mass: torch.Tensor, floor: float = 1e-7) -> torch.Tensor:
    """values / max-guarded mass (utils.py:135-142: masses <= 1e-7 are replaced by a float32 1e-7, which promotes)."""
    eps = torch.tensor(floor, dtype=torch.float32, device=mass.device)
    return values / torch.where(mass <= floor, eps, mass)


def _inverse_cdf(levels: torch.Tensor, cdf: torch.Tensor, support: torch.Tensor) -> torch.Tensor:
    """support[min(#{cdf < level}, n - 1)] per level (losses.py:214-220: left searchsorted, clamp, take_along_dim)."""
    rank = torch.searchsorted(cdf, levels)
    return torch.take_along_dim(support, rank.clamp(0, support.shape[1] - 1), dim=1)


def measures_to_levels(u_pos, v_pos, u_w, v_w, require_sort=True):
    """Sorted supports, CDFs, merged levels and both inverse CDFs on them (losses.py:286-298)."""
    if require_sort:
        u_pos, iu = torch.sort(u_pos, 1)
        v_pos, iv = torch.sort(v_pos, 1)
        u_w, v_w = torch.gather(u_w, 1, iu), torch.gather(v_w, 1, iv)
    cdf_u, cdf_v = torch.cumsum(u_w, 1), torch.cumsum(v_w, 1)
    levels = torch.sort(torch.cat((cdf_u, cdf_v), 1), 1)[0]
    return _inverse_cdf(levels, cdf_u, u_pos), _inverse_cdf(levels, cdf_v, v_pos), levels, cdf_u, cdf_v


def transport_rows(u_pos, v_pos, u_w, v_w, p=1, require_sort=True, return_quantiles=False, limit_quantile_range=False):
    """W_p^p per row of two discrete measures whose weights are used as given (losses.py:223-313)."""
    assert p >= 1, f"The OT loss is only valid for p>=1, {p} was given"
    uq, vq, levels, cdf_u, cdf_v = measures_to_levels(u_pos, v_pos, u_w, v_w, require_sort)
    if return_quantiles:
        return uq, vq, levels, cdf_u, cdf_v
    levels = torch.nn.functional.pad(levels, pad=(1, 0))
    widths = levels[..., 1:] - levels[..., :-1]
    if limit_quantile_range:   # levels beyond the first measure's total mass carry nothing (the paper's frequency cutoff)
        widths = torch.where(levels[..., 1:] > 1, torch.zeros_like(widths), widths)
    gap = torch.abs(uq - vq)
    return torch.sum(widths * (gap if p == 1 else gap.pow(p)), 1)


def module_forward(x, y, x_pos, y_pos, *, p, square_dist, dont_normalize, limit_quantile_range, require_sort, hinge_on,
                   hinge_value=0.0, dims=None, return_quantiles=False, rows_only=False):
    """Wasserstein1D.forward (losses.py:157-211) for tensors of any device / floating dtype."""
    lead = x.shape[:-1]
    flat = lambda t: t.reshape(-1, t.shape[-1]) if t.ndim == 3 else t   # noqa: E731
    x, y, x_pos, y_pos = flat(x), flat(y), flat(x_pos), flat(y_pos)
    if x_pos.ndim == 1:
        x_pos = x_pos.unsqueeze(0).expand_as(x)
    if y_pos.ndim == 1:
        y_pos = y_pos.unsqueeze(0).expand_as(y)
    if square_dist:
        x, y = x ** 2, y ** 2
    mass_x = torch.sum(x, dim=1, keepdim=True)
    a = guarded_ratio(x, mass_x)
    b = guarded_ratio(y, mass_x if dont_normalize else torch.sum(y, dim=1, keepdim=True))
    out = transport_rows(x_pos, y_pos, a, b, p=p, require_sort=require_sort, return_quantiles=return_quantiles,
                         limit_quantile_range=limit_quantile_range)
    if return_quantiles:
        return [t.reshape(lead + (-1,)) for t in out]
    if hinge_on:
        out = torch.nn.functional.relu(out - hinge_value)
    if rows_only:   # the flat per-row losses, before the reshape / mean (what the row-sharded multi-GPU reduction consumes)
        return out
    return torch.mean(out.reshape(lead), dim=dims)
