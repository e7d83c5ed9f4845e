"""oracle/torch_restatement.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Op-for-op PyTorch restatement of the reference's SOT loss, written from the
semantics in SURVEY.md Appendix A (reference: losses.py:129-313,
utils.py:135-142).  It issues the same ATen op sequence as the reference
(sort -> gather -> cumsum -> cat+sort -> searchsorted -> take_along_dim ->
pad/diff -> mask -> |.|^p -> sum -> mean), so that:
  * on CPU it is bit-identical to the imported reference (checked when the
    golden fixtures are generated, oracle/make_golden.py), and
  * bench.py can time "the reference's CPU PyTorch path" on the GPU box's
    host cores, where /root/reference does not exist (cpu_baseline.kind="port").
It also yields reference gradients through autograd for the backward tests.
Never imported by the product package.
"""
from __future__ import annotations

import torch


def _guarded(den: torch.Tensor) -> torch.Tensor:
    # utils.py:135-142: the epsilon is a float32 tensor whatever the input dtype
    eps = torch.tensor(1e-7, dtype=torch.float32, device=den.device)
    return torch.where(den <= 1e-7, eps, den)


def inverse_cdf(levels: torch.Tensor, cdf: torch.Tensor, support: torch.Tensor) -> torch.Tensor:
    """losses.py:214-220 (quantile_function): left rank, clamped, then lookup."""
    last = support.shape[1] - 1
    rank = torch.searchsorted(cdf, levels)
    return torch.take_along_dim(support, torch.clamp(rank, 0, last), dim=1)


def transport_cost_rows(u_values, v_values, u_weights=None, v_weights=None, p=1, require_sort=True,
                        return_quantiles=False, limit_quantile_range=False, stable_levels=False):
    """losses.py:223-313 (wasserstein_1d): W_p^p per row, no p-th root.

    stable_levels=True sorts the merged levels with a stable sort (the reference's default sort is
    unstable on this torch build): values are unchanged, only the routing of gradients between
    EQUAL levels differs -- the stable order (U before V, lower index first) is the tie convention
    the HIP backward and the C oracle implement."""
    assert p >= 1, f"The OT loss is only valid for p>=1, {p} was given"
    n, m = u_values.shape[1], v_values.shape[1]
    if u_weights is None:
        u_weights = torch.full(u_values.shape, 1.0 / n, device=u_values.device, dtype=u_values.dtype)
    if v_weights is None:
        v_weights = torch.full(v_values.shape, 1.0 / m, device=v_values.device, dtype=v_values.dtype)
    if require_sort:
        u_values, order_u = torch.sort(u_values, 1)
        v_values, order_v = torch.sort(v_values, 1)
        u_weights = torch.gather(u_weights, 1, order_u)
        v_weights = torch.gather(v_weights, 1, order_v)
    cdf_u = torch.cumsum(u_weights, 1)
    cdf_v = torch.cumsum(v_weights, 1)
    levels = torch.sort(torch.cat((cdf_u, cdf_v), 1), dim=1, stable=stable_levels)[0]
    q_u = inverse_cdf(levels, cdf_u, u_values)
    q_v = inverse_cdf(levels, cdf_v, v_values)
    if return_quantiles:
        return q_u, q_v, levels, cdf_u, cdf_v
    padded = torch.nn.functional.pad(levels, pad=(1, 0))
    width = padded[..., 1:] - padded[..., :-1]
    if limit_quantile_range:
        width = torch.where(padded[..., 1:] > 1, torch.zeros_like(width), width)
    gap = torch.abs(q_u - q_v)
    if p == 1:
        return torch.sum(width * gap, 1)
    return torch.sum(width * gap.pow(p), 1)


def sot_loss(x, y, x_pos, y_pos, p=1, square_dist=False, dont_normalize=False, limit_quantile_range=False,
             require_sort=True, hinge=False, hinge_value=0.0, dims=None, return_quantiles=False,
             reduce=True, stable_levels=False):
    """losses.py:129-211 (Wasserstein1D.forward) with the ctor/call flags flattened."""
    lead = x.shape[:-1]
    if x.ndim == 3:
        x = x.reshape(-1, x.shape[-1])
    if y.ndim == 3:
        y = y.reshape(-1, y.shape[-1])
    if x_pos.ndim == 3:
        x_pos = x_pos.reshape(-1, x_pos.shape[-1])
    if y_pos.ndim == 3:
        y_pos = y_pos.reshape(-1, y_pos.shape[-1])
    if x_pos.ndim == 1:
        x_pos = x_pos.unsqueeze(0).expand_as(x)
    if y_pos.ndim == 1:
        y_pos = y_pos.unsqueeze(0).expand_as(y)
    if square_dist:
        x = x ** 2
        y = y ** 2
    mass_x = torch.sum(x, dim=1, keepdim=True)
    x = x / _guarded(mass_x)
    if dont_normalize:
        y = y / _guarded(mass_x)
    else:
        y = y / _guarded(torch.sum(y, dim=1, keepdim=True))
    rows = transport_cost_rows(x_pos, y_pos, u_weights=x, v_weights=y, p=p, require_sort=require_sort,
                               return_quantiles=return_quantiles, limit_quantile_range=limit_quantile_range,
                               stable_levels=stable_levels)
    if return_quantiles:
        return [t.reshape(lead + (-1,)) for t in rows]
    if hinge:
        rows = torch.nn.functional.relu(rows - hinge_value)
    rows = rows.reshape(lead)
    if not reduce:
        return rows
    return torch.mean(rows, dim=dims)
