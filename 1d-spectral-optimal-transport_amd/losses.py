"""Host-side mirror of the reference's loss interface for the SOT hot path.

Same names, argument meaning and error behaviour as the reference's ``losses`` module
(/root/reference/losses.py): ``Wasserstein1D`` (:89-211), ``quantile_function`` (:214-220),
``wasserstein_1d`` (:223-313), ``Wasserstein1DWithTransform`` (:316-343), ``MixOfLosses`` (:346-362), ``MSSLoss``
(:365-425) and ``utils.safe_divide``
(utils.py:135-142), so ``trainer.py:220/228``, ``metrics.py:148`` and the YAML
``class_path: losses.Wasserstein1D`` keep working unchanged.

All arithmetic on float32 HIP-device tensors runs in the hand-written HIP library (csrc/, C ABI in
include/sot_hip.h); this file only reshapes, marshals pointers and wires autograd.  Tensors outside the library's domain
-- CPU tensors, float64 -- take the reference's own route, torch ops (_torch_path.py), because the reference is device- and
dtype-agnostic (losses.py:129-211) and `accelerator: cpu` has to keep working.
"""
from __future__ import annotations

import typing
import warnings

import torch

from . import _native as nat
from . import _torch_path as tp

# (wasserstein_1d_csr -- ragged supports in CSR form -- is retired from the public surface since round 4: zero-masked dense rows are
#  config 4; the function itself stays for its tests and the C entry point)
__all__ = ["Wasserstein1D", "Wasserstein1DWithTransform", "wasserstein_1d", "quantile_function", "MixOfLosses", "MSSLoss", "safe_divide"]

FLAG_PRENORMALIZED = nat.FLAG_PRENORMALIZED


def safe_divide(numerator, denominator, eps=1e-7):
    """utils.py:135-142 -- kept for API compatibility (the fused kernel applies the same guard)."""
    safe_denominator = torch.where(denominator <= eps,
                                   torch.tensor(eps, dtype=torch.float32, device=denominator.device), denominator)
    return numerator / safe_denominator


_warned = set()


def warn_once(key, message):
    """One warning per distinct situation (e.g. per unsupported transform size), not one per call."""
    if key not in _warned:
        _warned.add(key)
        warnings.warn(message, stacklevel=3)


def _hip_domain(*tensors) -> bool:
    """True when the HIP library takes these tensors: all of them float32 on a HIP device.  CPU tensors (any floating dtype) and
    float64 tensors (any device) go to the torch-op composition instead -- float64 is then float64 arithmetic, as in the
    reference; a float64 GPU tensor says so once.  half / bfloat16 GPU tensors are refused (TypeError), not upcast."""
    ts = [t for t in tensors if t is not None]
    if all(t.is_cuda and t.dtype == torch.float32 for t in ts):
        return True
    for t in ts:
        if t.is_cuda and t.dtype in (torch.float16, torch.bfloat16):
            raise TypeError(f"sot_amd computes in float32; got {t.dtype} (convert with .float())")
    if any(t.is_cuda for t in ts):
        warn_once("not-hip-f32", "sot_amd: tensors that are not all float32 on the GPU (float64, or CPU and GPU mixed) run the torch-op "
                                 "composition in their own dtype, like the reference; the HIP kernels take float32 GPU tensors")
    return False


ROW_SIDE_LIMIT = 16384    # points of ONE side (csrc/sot_hip.hip pick_cfg: at most 1024 threads x 16 points per row)
ROW_POINT_LIMIT = 18000   # n + m of the longest row pair the kernels take for sure (one pair's working set lives in ONE CU's 160 KiB
                          # of LDS; include/sot_hip.h: SOT_ERR_UNSUPPORTED_SIZE beyond ~19000 forward / ~13000 backward)


def _beyond_one_cu(x, y, *positions) -> bool:
    """Rows too long for the LDS-resident kernels (n_fft >= 32768): the reference has no size limit (losses.py:223-313), so such GPU
    tensors run the package's torch-op composition -- said once -- instead of failing.  No paper configuration comes near."""
    n, m = x.shape[-1], y.shape[-1]
    if any(t is not None and t.ndim >= 2 and not (t.shape[0] > 1 and t.stride(0) == 0) for t in positions):
        # per-row positions are sorted in LDS on power-of-two arrays (csrc/sot_hip.hip make_layout, rowpos)
        n, m = 1 << max(n - 1, 0).bit_length(), 1 << max(m - 1, 0).bit_length()
    wants_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, y) + tuple(positions))
    limit = ROW_POINT_LIMIT if not wants_grad else 12000
    if n + m <= limit and max(n, m) <= ROW_SIDE_LIMIT:   # the kernels also cap each side (pick_cfg: 1024 threads x 16 points)
        return False
    warn_once("row-too-long", f"sot_amd: rows of {n} + {m} points exceed what one CU's LDS holds; this call runs the torch-op composition "
                              "on the GPU instead of the HIP kernels")
    return True


def _fresh_grid_ok(pos, x, width) -> bool:
    """A caller-made position tensor the fresh-positions branch of Wasserstein1D.forward can hand to the C++ host path as it is."""
    return (pos.ndim == 1 and pos.shape[0] == width and pos.dtype is torch.float32 and pos.is_cuda and pos.device == x.device
            and pos.stride(0) == 1)


def _flags(square_dist, dont_normalize, limit_quantile_range, require_sort, prenormalized=False):
    return ((nat.FLAG_SQUARE if square_dist else 0) | (nat.FLAG_DONT_NORMALIZE if dont_normalize else 0)
            | (nat.FLAG_LIMIT_Q if limit_quantile_range else 0) | (nat.FLAG_REQUIRE_SORT if require_sort else 0)
            | (FLAG_PRENORMALIZED if prenormalized else 0))


def _position_grads(x, y, xpos, ypos, p, flags, plan, grad_rows, need_x, need_y, perm_in=None):
    """d sum_r grad_rows[r] * loss_r / d positions (losses.py:287-313: the positions enter through torch.sort and
    take_along_dim, both differentiable).  No reference call site asks for it; autograd supplies it, and so does
    this package: ONE deterministic HIP kernel (sot_w1d_position_grad: the merge walk accumulates
        d loss / d xs[i] = sum over the merged levels k whose x-rank is i of  delta_k * p |uq_k - vq_k|^(p-1) sign(uq_k - vq_k)
    and minus that for ys, no atomics) plus the fixed-order batch sum for a position row shared by all rows (sot_column_sum)."""
    try:
        return nat.position_grads(x, y, xpos, ypos, p, flags, grad_rows, need_x, need_y, plan, perm_in=perm_in)
    except nat.SotError as err:   # rows whose gradient layout + per-thread tails exceed one CU's LDS (n + m beyond ~12 000)
        if err.status != nat.SOT_ERR_UNSUPPORTED_SIZE:
            raise
        warn_once("posgrad-too-long", "sot_amd: position gradients of rows this long run as torch ops (atomic adds)")
        return _position_grads_torch(x, y, xpos, ypos, p, flags, plan, grad_rows, need_x, need_y)


def _position_grads_torch(x, y, xpos, ypos, p, flags, plan, grad_rows, need_x, need_y):
    """The same gradient from the quantile kernel's outputs with torch ops (searchsorted + scatter_add_): only for rows beyond the
    position-gradient kernel's LDS budget."""
    uq, vq, levels, cdf_x, cdf_y = nat.quantiles(x, y, xpos, ypos, p, flags, plan)
    rows, n = x.shape
    m = y.shape[1]
    delta = torch.diff(levels, dim=1, prepend=torch.zeros(rows, 1, device=x.device))
    if flags & nat.FLAG_LIMIT_Q:
        delta = torch.where(levels > 1, torch.zeros_like(delta), delta)
    d = uq - vq
    slope = torch.sign(d) if p == 1 else p * d.abs().pow(p - 1) * torch.sign(d)
    w = delta * slope * grad_rows.reshape(-1, 1)
    out = []
    for need, pos, cdf, width, sgn in ((need_x, xpos, cdf_x, n, 1.0), (need_y, ypos, cdf_y, m, -1.0)):
        if not need:
            out.append(None)
            continue
        rank = torch.searchsorted(cdf, levels).clamp_(max=width - 1)          # quantile_function, losses.py:214-220
        g_sorted = torch.zeros(rows, width, device=x.device).scatter_add_(1, rank, sgn * w)
        pos2 = pos if pos.ndim == 2 else pos.unsqueeze(0)
        if flags & nat.FLAG_REQUIRE_SORT:                                       # back through torch.sort (losses.py:286-288)
            order = torch.sort(pos2, dim=1)[1].expand(rows, width)
            g = torch.zeros_like(g_sorted).scatter_(1, order, g_sorted)
        else:
            g = g_sorted
        out.append(g.sum(0) if pos.ndim == 1 else g)
    return out


class _RowLoss(torch.autograd.Function):
    """rows[B] = W_p^p per spectrum pair; backward = closed-form HIP kernel.  Per-row positions (the `torch.sort(u_values, 1)`
    of losses.py:286-288 on every row) are sorted ONCE per step: the forward leaves each row's permutations in a uint16 buffer and the
    backward / position-gradient kernels gather through them (sot_problem.row_perm_out / row_perm_in)."""

    @staticmethod
    def forward(ctx, x, y, xpos, ypos, p, flags, plan):
        perm = nat.row_permutations(x, y, xpos, ypos, flags) if (plan is None and any(ctx.needs_input_grad[:4])) else None
        rows = nat.forward_rows(x, y, xpos, ypos, p, flags, plan, perm_out=perm)
        ctx.save_for_backward(x, y, xpos, ypos, *([perm] if perm is not None else []))
        ctx.p, ctx.flags, ctx.plan = p, flags, plan
        return rows

    @staticmethod
    def backward(ctx, grad_rows):
        x, y, xpos, ypos, *perm = ctx.saved_tensors
        perm = perm[0] if perm else None
        gx = gy = gxp = gyp = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx, gy = nat.backward_rows(x, y, xpos, ypos, ctx.p, ctx.flags, grad_rows.float(),
                                       need_gx=ctx.needs_input_grad[0], need_gy=ctx.needs_input_grad[1], plan=ctx.plan, perm_in=perm)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            gxp, gyp = _position_grads(x, y, xpos, ypos, ctx.p, ctx.flags, ctx.plan, grad_rows.float(), ctx.needs_input_grad[2],
                                       ctx.needs_input_grad[3], perm_in=perm)
        return gx, gy, gxp, gyp, None, None, None


EARLY_GRADIENT = True   # module switch for the loss-and-gradient form below (tests compare both forms)


class _FusedMeanLoss(torch.autograd.Function):
    """losses.py:129-211 with dims=None behind ONE native call (forward kernel + fixed-order batch-mean kernel); the backward feeds the
    scalar upstream gradient to the HIP backward kernel as a broadcast value scaled by 1/B.

    When only y needs a gradient (the training step: x is the target spectrum) the gradient of the mean is computed
    together with the loss (sot_w1d_loss_and_grad: one pass over the rows instead of forward + backward) and only rescaled
    by the upstream scalar in backward (a no-op kernel for a plain ``loss.backward()``)."""

    @staticmethod
    def forward(ctx, x, y, xpos, ypos, p, flags, plan):
        ctx.p, ctx.flags, ctx.plan = p, flags, plan
        ctx.early_gy = None
        if ctx.needs_input_grad[1] and not ctx.needs_input_grad[0] and EARLY_GRADIENT:
            mean, _, ctx.early_gy = nat.loss_and_grad(x, y, xpos, ypos, p, flags, plan)
        else:
            mean, _, _ = nat.loss_fused(x, y, xpos, ypos, p, flags, plan)
        ctx.save_for_backward(x, y, xpos, ypos)
        return mean

    @staticmethod
    def backward(ctx, g):
        x, y, xpos, ypos = ctx.saved_tensors
        gxp = gyp = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            gxp, gyp = _position_grads(x, y, xpos, ypos, ctx.p, ctx.flags, ctx.plan, (g.float() / x.shape[0]).expand(x.shape[0]),
                                       ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        if ctx.early_gy is not None:   # consumed once: a second backward through a retained graph recomputes below
            gy, ctx.early_gy = ctx.early_gy, None
            return None, nat.scale_inplace(gy, g.float().contiguous()), gxp, gyp, None, None, None
        gx = gy = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx, gy = nat.backward_rows(x, y, xpos, ypos, ctx.p, ctx.flags, g.float(), need_gx=ctx.needs_input_grad[0],
                                       need_gy=ctx.needs_input_grad[1], plan=ctx.plan, grad_scale=1.0 / x.shape[0])
        return gx, gy, gxp, gyp, None, None, None


class _RowMean(torch.autograd.Function):
    """losses.py:211 with dims=None: fixed-order fp64 accumulation on the GPU."""

    @staticmethod
    def forward(ctx, rows):
        ctx.count = rows.numel()
        return nat.reduce_mean(rows)

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.count).expand(ctx.count)


def _version_of(t):
    try:
        return t._version
    except RuntimeError:  # inference tensors (metrics.py:144 runs under inference_mode) are immutable
        return -1


class _PlanCache:
    """Position plans keyed by tensor identity+version, so a persistent grid is sorted once."""

    def __init__(self, capacity=4):
        self.capacity = capacity
        self.entries = []  # (key, xpos_ref, ypos_ref, plan)

    def get(self, xpos, ypos):
        key = (xpos.data_ptr(), ypos.data_ptr(), xpos.numel(), ypos.numel(), _version_of(xpos), _version_of(ypos),
               xpos.device)
        for k, _, _, plan in self.entries:
            if k == key:
                return plan
        plan = nat.PositionPlan(xpos.detach(), ypos.detach())
        self.entries.append((key, xpos, ypos, plan))  # tensor refs pin the storage behind data_ptr
        if len(self.entries) > self.capacity:
            self.entries.pop(0)
        return plan


def _prepare(x, y, xpos, ypos, checked=False):
    """Common marshalling: 2-D fp32 HIP rows and matching positions (checked: the caller has established _hip_domain)."""
    if not checked:
        nat.require_hip(x, y, xpos, ypos)
    if x.ndim != 2 or y.ndim != 2:
        raise ValueError(f"expected 2-D [rows, features] weights, got {tuple(x.shape)} and {tuple(y.shape)}")
    if x.shape[0] != y.shape[0]:
        raise RuntimeError(f"row count mismatch: {x.shape[0]} vs {y.shape[0]}")
    x, y = nat.rows_view(x), nat.rows_view(y)
    if xpos.shape[-1] != x.shape[1] or ypos.shape[-1] != y.shape[1]:
        raise RuntimeError("positions and weights must have the same number of features")
    if (xpos.ndim == 1) != (ypos.ndim == 1):  # mixed: materialise the shared row for every batch row
        if xpos.ndim == 1:
            xpos = xpos.unsqueeze(0).expand_as(x)
        else:
            ypos = ypos.unsqueeze(0).expand_as(y)
    if xpos.ndim == 2:
        if xpos.shape[0] != x.shape[0] or ypos.shape[0] != y.shape[0]:
            raise RuntimeError("per-row positions must have one row per weight row")
        xpos, ypos = xpos.contiguous(), ypos.contiguous()
    else:
        xpos, ypos = xpos.contiguous(), ypos.contiguous()
    return x, y, xpos, ypos


_functional_plans = _PlanCache()


def quantile_function(qs, cws, xs):
    """losses.py:214-220.  Compatibility helper only: inside this package the inverse-CDF lookup is
    fused into the merge kernel and this function is never called on the hot path."""
    n = xs.shape[1]
    idx = torch.searchsorted(cws, qs)
    return torch.take_along_dim(xs, torch.clamp(idx, 0, n - 1), dim=1)


def wasserstein_1d(u_values, v_values, u_weights=None, v_weights=None, p=1, require_sort=True,
                   return_quantiles=False, limit_quantile_range=False):
    """losses.py:223-313: W_p^p per row of two discrete measures (weights used as given)."""
    assert p >= 1, f"The OT loss is only valid for p>=1, {p} was given"
    n, m = u_values.shape[1], v_values.shape[1]
    if u_weights is None:
        u_weights = torch.full(u_values.shape, 1.0 / n, device=u_values.device, dtype=u_values.dtype)
    if v_weights is None:
        v_weights = torch.full(v_values.shape, 1.0 / m, device=v_values.device, dtype=v_values.dtype)
    if not _hip_domain(u_values, v_values, u_weights, v_weights) or _beyond_one_cu(u_weights, v_weights, u_values, v_values):
        return tp.transport_rows(u_values, v_values, u_weights, v_weights, p=p, require_sort=require_sort,
                                 return_quantiles=return_quantiles, limit_quantile_range=limit_quantile_range)

    # stride-0 expanded positions (losses.py:167-170) are passed down as one shared row
    def shared_row(t):
        return t[0] if (t.ndim == 2 and t.shape[0] > 1 and t.stride(0) == 0) else t
    upos, vpos = shared_row(u_values), shared_row(v_values)
    x, y, upos, vpos = _prepare(u_weights, v_weights, upos, vpos)
    flags = _flags(False, False, limit_quantile_range, require_sort, prenormalized=True)
    plan = _functional_plans.get(upos, vpos) if (require_sort and upos.ndim == 1) else None
    if return_quantiles:
        return nat.quantiles(x, y, upos, vpos, p, flags, plan)
    return _RowLoss.apply(x, y, upos, vpos, float(p), flags, plan)


def _csr_to_padded(weights, positions, offsets, width):
    """CSR entries -> [rows, width] weights (zeros behind each row's entries: zero-weight points are inert) and positions (the
    row's last position repeated), plus the mask of real entries (row-major = CSR order)."""
    starts, ends = offsets[:-1, None], offsets[1:, None]
    idx = starts + torch.arange(width, device=offsets.device)[None, :]
    valid = idx < ends
    idx = torch.minimum(idx, (ends - 1).clamp_min(0)).clamp_(0, max(weights.numel() - 1, 0))
    return torch.where(valid, weights[idx], torch.zeros((), device=weights.device)), positions[idx], valid


class _CsrRowLoss(torch.autograd.Function):
    """Row losses of the CSR form (sot_w1d_forward_csr).  Backward: the rows are padded with zero-weight points to
    [rows, max_n] / [rows, max_m] with per-row positions and go through the dense backward kernel (the same function of the kept
    weights, hence the same gradient); the gradients of the kept points are gathered back into CSR order."""

    @staticmethod
    def forward(ctx, xw, xp, xoff, yw, yp, yoff, max_n, max_m, p, flags):
        rows = nat.forward_rows_csr(xw, xp, xoff, yw, yp, yoff, max_n, max_m, p, flags)
        ctx.save_for_backward(xw, xp, xoff, yw, yp, yoff)
        ctx.cfg = (max_n, max_m, p, flags)
        return rows

    @staticmethod
    def backward(ctx, grad_rows):
        xw, xp, xoff, yw, yp, yoff = ctx.saved_tensors
        max_n, max_m, p, flags = ctx.cfg
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[4]:
            raise NotImplementedError("gradients w.r.t. support positions are not implemented (no reference call site uses them)")
        xd, xpd, xvalid = _csr_to_padded(xw, xp, xoff, max_n)
        yd, ypd, yvalid = _csr_to_padded(yw, yp, yoff, max_m)
        gx, gy = nat.backward_rows(xd, yd, xpd.contiguous(), ypd.contiguous(), p, flags, grad_rows.float(),
                                   need_gx=ctx.needs_input_grad[0], need_gy=ctx.needs_input_grad[3])
        return (gx[xvalid] if gx is not None else None, None, None, gy[yvalid] if gy is not None else None, None, None,
                None, None, None, None)


def wasserstein_1d_csr(x_weights, x_positions, x_offsets, y_weights, y_positions, y_offsets, max_n, max_m, p=1,
                       square_dist=False, dont_normalize=False, limit_quantile_range=False, require_sort=True,
                       prenormalized=False):
    """Per-row SOT loss for RAGGED supports in CSR form (no reference counterpart: the reference is fed zero-masked
    dense rows, which give the same value because zero-weight points are inert -- BASELINE config 4).
    Row r owns entries [offsets[r], offsets[r+1]); normalisation etc. follow Wasserstein1D's keyword arguments.
    Returns the [rows] tensor of W_p^p; differentiable w.r.t. the weights (see _CsrRowLoss)."""
    assert p >= 1, f"The OT loss is only valid for p>=1, {p} was given"
    flags = _flags(square_dist, dont_normalize, limit_quantile_range, require_sort, prenormalized)
    if torch.is_grad_enabled() and (x_weights.requires_grad or y_weights.requires_grad):
        return _CsrRowLoss.apply(x_weights, x_positions, x_offsets, y_weights, y_positions, y_offsets, int(max_n), int(max_m),
                                 float(p), flags)
    return nat.forward_rows_csr(x_weights, x_positions, x_offsets, y_weights, y_positions, y_offsets, max_n, max_m,
                                float(p), flags)


class _HotCall(typing.NamedTuple):
    """What Wasserstein1D.forward remembers of its last default call (round-5 review: a 13-slot positional tuple before)."""
    x_pos: torch.Tensor          # the position tensors of that call, by identity ...
    y_pos: torch.Tensor
    x_version: int               # ... and their in-place-modification counters then
    y_version: int
    settings: tuple              # Wasserstein1D._settings() then
    plan: object                 # the position plan (sorted grids, permutations, identity flags)
    flag_word: int               # the call's flag word as the library takes it (with SOT_FLAG_SAME_GRID where the plan says so)
    glue: object                 # the C++ host path (_sot_glue)
    n: int                       # row lengths
    m: int
    fresh_ok: bool               # the fresh-positions branch applies: caller-passed positions (not the fixed_x buffer) and the extension has mean_loss_fresh
    flags_no_same_grid: int      # the flag word without the same-grid bit (two fresh tensors: the C++ side puts it back when both are one tensor)
    area_ok: bool                # the merge-free kernel may be asked for (p = 1, no cutoff)


class Wasserstein1D(torch.nn.Module):
    """Drop-in for losses.Wasserstein1D (losses.py:89-211); see that docstring for the arguments."""

    def __init__(self, p=1, fixed_x=None, require_sort=True, log_scaled_x=False, **kwargs):
        super().__init__()
        self.p = p
        self.require_sort = require_sort
        self.log_scaled_x = log_scaled_x  # flag only, like the reference (trainer.py:187 reads it)
        self.dont_normalize = kwargs.get("dont_normalize", False)
        self.limit_quantile_range = kwargs.get("limit_quantile_range", False)
        self.hinge = kwargs.get("hinge", False)
        self.square_dist = kwargs.get("square_dist", False)
        # NOT a reference argument (default: the reference's behaviour).  True: for p = 1 on one grid the gradient w.r.t. y comes from the
        # merge-free training form -- the derivative of the loss in the CDF values, which is what float64 autograd of the reference
        # returns; at exactly tied float32 levels the reference's float32 autograd returns a tie-order artefact instead
        self.tie_free_gradient = bool(kwargs.get("tie_free_gradient", False))
        # unknown kwargs (e.g. cumsum_only from the paper YAMLs) are accepted and ignored, losses.py:96
        if fixed_x is not None:
            self.register_buffer("fixed_x", torch.linspace(0, 1, fixed_x))
        else:
            self.register_buffer("fixed_x", None)
        self._plans = _PlanCache()
        self._hot = None   # the last default call, a _HotCall: see forward

    def __getstate__(self):
        # the caches hold device events and the extension module: a copy / pickle of the module (copy.deepcopy for an EMA model,
        # torch.save(model)) starts with empty ones
        state = self.__dict__.copy()
        state["_plans"] = None
        state["_hot"] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plans = _PlanCache()
        self._hot = None

    def _positions(self, x_pos, y_pos):
        if (x_pos is None or y_pos is None) and self.fixed_x is None:
            raise ValueError("If fixed_x is not provided, x_pos and y_pos must be provided")
        assert self.p >= 1, f"The OT loss is only valid for p>=1, {self.p} was given"  # losses.py:271
        return (self.fixed_x if x_pos is None else x_pos), (self.fixed_x if y_pos is None else y_pos)

    def _marshal(self, x, y, x_pos, y_pos, kwargs):
        """float32 GPU tensors -> what the native calls take: 2-D rows, positions, flag word, position plan."""
        x_pos_, y_pos_ = self._positions(x_pos, y_pos)
        original_shape = x.shape[:-1]
        if x.ndim == 3:
            x = x.reshape(-1, x.shape[-1])
        if y.ndim == 3:
            y = y.reshape(-1, y.shape[-1])
        if x_pos_.ndim == 3:
            x_pos_ = x_pos_.reshape(-1, x_pos_.shape[-1])
        if y_pos_.ndim == 3:
            y_pos_ = y_pos_.reshape(-1, y_pos_.shape[-1])

        dont_normalize = bool(kwargs.get("dont_normalize", False) or self.dont_normalize)
        limit_q = bool(kwargs.get("limit_quantile_range", False) or self.limit_quantile_range)
        flags = _flags(self.square_dist, dont_normalize, limit_q, self.require_sort)
        if getattr(self, "tie_free_gradient", False):
            flags |= nat.FLAG_TIE_FREE_GRADIENT

        x, y, x_pos_, y_pos_ = _prepare(x, y, x_pos_, y_pos_, checked=True)
        plan = self._plans.get(x_pos_, y_pos_) if (self.require_sort and x_pos_.ndim == 1) else None
        return x, y, x_pos_, y_pos_, flags, plan, original_shape

    def _torch_forward(self, x, y, x_pos_, y_pos_, kwargs, rows_only=False):
        """CPU / float64 tensors: the torch-op composition (_torch_path.py), same keyword handling as the HIP route."""
        return tp.module_forward(
            x, y, x_pos_, y_pos_, p=self.p, square_dist=self.square_dist,
            dont_normalize=bool(kwargs.get("dont_normalize", False) or self.dont_normalize),
            limit_quantile_range=bool(kwargs.get("limit_quantile_range", False) or self.limit_quantile_range),
            require_sort=self.require_sort, hinge_on=bool(self.hinge), hinge_value=kwargs.get("hinge", 0.0),
            dims=(0 if rows_only else kwargs.get("dims", None)), return_quantiles=(not rows_only) and kwargs.get("return_quantiles", False),
            rows_only=rows_only)

    def row_losses(self, x, y, x_pos=None, y_pos=None, **kwargs):
        """Flat [rows] tensor of W_p^p per spectrum pair (after the optional hinge, before the mean):
        what losses.py:186-205 holds before its reshape/mean.  Used by the row-sharded multi-GPU path."""
        x_pos_, y_pos_ = self._positions(x_pos, y_pos)
        if not _hip_domain(x, y, x_pos_, y_pos_) or _beyond_one_cu(x, y, x_pos_, y_pos_):
            return self._torch_forward(x, y, x_pos_, y_pos_, kwargs, rows_only=True)
        x, y, x_pos_, y_pos_, flags, plan, _ = self._marshal(x, y, x_pos, y_pos, kwargs)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (x, y, x_pos_, y_pos_)):
            loss = _RowLoss.apply(x, y, x_pos_, y_pos_, float(self.p), flags, plan)
        else:
            loss = nat.forward_rows(x, y, x_pos_, y_pos_, float(self.p), flags, plan)
        if self.hinge:
            loss = torch.nn.functional.relu(loss - kwargs.get("hinge", 0.0))
        return loss

    def _settings(self):
        return (self.p, self.square_dist, self.dont_normalize, self.limit_quantile_range, self.require_sort, self.hinge,
                getattr(self, "tie_free_gradient", False))

    def forward(self, x, y, x_pos=None, y_pos=None, **kwargs):
        # The hot call, recognised before anything else is looked at: the SAME position tensors as last time (a persistent grid or the
        # fixed_x buffer), no call keywords, float32 GPU rows -> straight into the C++ host path with the cached plan and flag word.
        # Everything the general path below would re-derive per call (domain check, reshapes of positions, flags, plan lookup) was
        # derived when the entry was made; the C++ side checks the weight tensors themselves.
        hot = self._hot
        if (hot is not None and EARLY_GRADIENT and not kwargs and x.dtype is torch.float32 and y.dtype is torch.float32 and x.is_cuda
                and y.is_cuda):
            xp = self.fixed_x if x_pos is None else x_pos
            yp = self.fixed_x if y_pos is None else y_pos
            if (xp is hot.x_pos and yp is hot.y_pos and _version_of(xp) == hot.x_version and _version_of(yp) == hot.y_version and self._settings() == hot.settings
                    and 2 <= x.ndim <= 3 and x.ndim == y.ndim):
                x2 = x if x.ndim == 2 else x.reshape(-1, x.shape[-1])
                y2 = y if y.ndim == 2 else y.reshape(-1, y.shape[-1])
                if (x2.stride(1) == 1 and y2.stride(1) == 1 and x2.shape[1] == hot.n and y2.shape[1] == hot.m
                        and not (torch.is_grad_enabled() and (x2.requires_grad or xp.requires_grad or yp.requires_grad))):
                    plan = hot.plan
                    plan.use_on_current_stream(x2.device)
                    return hot.glue.mean_loss(x2, y2, plan.xpos_sorted, plan.ypos_sorted, plan.xperm, plan.yperm, plan.ident, float(self.p), hot.flag_word)
            elif (x_pos is not None and y_pos is not None and self._settings() == hot.settings and 2 <= x.ndim <= 3 and x.ndim == y.ndim
                  and hot.fresh_ok and _fresh_grid_ok(xp, x, hot.n) and _fresh_grid_ok(yp, x, hot.m)):
                # FRESH position tensors of the shapes seen last time -- the reference's trainer rebuilds its grid on every step
                # (trainer.py:187-197: x_pos = torch.tensor(freqs).to(device) / max, y_pos = x_pos.clone()).  Their content cannot be
                # compared on the host without a synchronisation, so the call gets its own plan, but inside the SAME C++ call as the loss
                # (one allocation, one two-workgroup launch in front of the row kernel): no Python plan object, no cache bookkeeping, no
                # general path.  flags_no_same_grid: the flag word WITHOUT the same-grid bit (whether two fresh tensors hold one grid is device-side
                # knowledge; the C++ side puts it back when both arguments are one tensor).
                x2 = x if x.ndim == 2 else x.reshape(-1, x.shape[-1])
                y2 = y if y.ndim == 2 else y.reshape(-1, y.shape[-1])
                if (x2.stride(1) == 1 and y2.stride(1) == 1 and x2.shape[1] == hot.n and y2.shape[1] == hot.m
                        and not (torch.is_grad_enabled() and (x2.requires_grad or xp.requires_grad or yp.requires_grad))):
                    return hot.glue.mean_loss_fresh(x2, y2, xp, yp, float(self.p), hot.flags_no_same_grid | (nat.FLAG_SAME_GRID if (xp is yp and hot.area_ok) else 0))
        x_pos_, y_pos_ = self._positions(x_pos, y_pos)
        if not _hip_domain(x, y, x_pos_, y_pos_) or _beyond_one_cu(x, y, x_pos_, y_pos_):
            return self._torch_forward(x, y, x_pos_, y_pos_, kwargs)
        return self._hip_forward(x, y, x_pos, y_pos, x_pos_, y_pos_, kwargs)

    def _hip_forward(self, x, y, x_pos, y_pos, x_pos_, y_pos_, kwargs):
        if kwargs.get("return_quantiles", False):
            x2, y2, x_pos_, y_pos_, flags, plan, original_shape = self._marshal(x, y, x_pos, y_pos, kwargs)
            out = nat.quantiles(x2, y2, x_pos_, y_pos_, float(self.p), flags, plan)
            return [t.reshape(original_shape + (-1,)) for t in out]

        original_shape = x.shape[:-1]
        dims = kwargs.get("dims", None)
        if dims is None and not self.hinge:
            # default reduction: forward and the mean over every row (losses.py:211) in one native call
            x2, y2, x_pos_, y_pos_, flags, plan, _ = self._marshal(x, y, x_pos, y_pos, kwargs)
            grad_on = torch.is_grad_enabled()
            glue = nat.glue() if (plan is not None and EARLY_GRADIENT) else None
            if glue is not None and not (grad_on and (x2.requires_grad or x_pos_.requires_grad or y_pos_.requires_grad)):
                # the hot call (trainer.py:220-228: gradient for the estimate's spectrum only; metrics.py:148: none): C++ host path
                plan.use_on_current_stream(x2.device)
                fl = nat.problem_flags(self.p, flags, plan)
                # remember the hot call (see forward) -- not from inside a stream capture, where plan.same_grid() answers "no" without
                # remembering it and the flag word frozen here would lack SOT_FLAG_SAME_GRID for good
                if (not kwargs and x_pos_.ndim == 1 and y_pos_.ndim == 1 and x2.shape[1] + y2.shape[1] <= 12000
                        and not torch.cuda.is_current_stream_capturing()):
                    self._hot = _HotCall(
                        x_pos=x_pos_, y_pos=y_pos_, x_version=_version_of(x_pos_), y_version=_version_of(y_pos_), settings=self._settings(), plan=plan,
                        flag_word=fl, glue=glue, n=x2.shape[1], m=y2.shape[1],
                        fresh_ok=x_pos is not None and y_pos is not None and hasattr(glue, "mean_loss_fresh"), flags_no_same_grid=int(flags),
                        area_ok=self.p == 1 and not (flags & (nat.FLAG_LIMIT_Q | nat.FLAG_NO_AREA | nat.FLAG_PRENORMALIZED)))
                return glue.mean_loss(x2, y2, plan.xpos_sorted, plan.ypos_sorted, plan.xperm, plan.yperm, plan.ident, float(self.p), fl)
            if grad_on and any(t.requires_grad for t in (x2, y2, x_pos_, y_pos_)):
                return _FusedMeanLoss.apply(x2, y2, x_pos_, y_pos_, float(self.p), flags, plan)
            return nat.loss_fused(x2, y2, x_pos_, y_pos_, float(self.p), flags, plan)[0]
        loss = self.row_losses(x, y, x_pos, y_pos, **kwargs)
        if dims is None:
            return _RowMean.apply(loss)  # torch.mean over every row -> 0-d tensor (losses.py:211)
        loss = loss.reshape(original_shape)
        return torch.mean(loss, dim=dims)


class _MultiScaleSpectral(torch.autograd.Function):
    """MSSLoss behind ONE autograd node: per FFT size two STFT-magnitude kernels and the spectral-distance kernels
    (include/sot_hip.h: sot_stft_mag_*, sot_spec_distance_*); the backward walks the scales again (distance backward ->
    STFT backward) and accumulates the audio gradients."""

    @staticmethod
    def forward(ctx, target_audio, audio, fft_sizes, mag_weight, logmag_weight, l2, per_item=False):
        from . import spectra
        ctx.per_item = per_item   # one value per clip (`dims` = the two spectrogram axes) instead of the scalar
        target_audio, audio = target_audio.contiguous(), audio.contiguous()
        total = None
        saved, specs = [], []
        for size in fft_sizes:
            hop = int(size * (1.0 - 0.75))                      # compute_mag's default overlap (features.py:214-216)
            win = spectra._cached_window(None, size, audio.device)   # window=None -> hann (features.py:203-204)
            cplx = None
            if target_audio.shape == audio.shape:
                if ctx.needs_input_grad[1] and spectra.SAVE_SPECTRUM:   # + the estimate's complex spectrum for the backward
                    t, v, cplx = nat.stft_mag_forward_pair(target_audio, audio, win, size, hop, want_spec_b=True)
                else:
                    t, v = nat.stft_mag_forward_pair(target_audio, audio, win, size, hop)   # one launch for both signals
            else:
                t, v = nat.stft_mag_forward(target_audio, win, size, hop), nat.stft_mag_forward(audio, win, size, hop)
            total = nat.spec_distance_forward(t, v, mag_weight, logmag_weight, 1e-5, l2, accumulate_into=total, per_row=per_item)   # total += d
            saved += [t, v]
            specs.append(cplx)
        have = [c is not None for c in specs]
        ctx.save_for_backward(target_audio, audio, *saved, *[c for c in specs if c is not None])
        ctx.cfg = (tuple(fft_sizes), mag_weight, logmag_weight, l2, have)
        return total

    @staticmethod
    def backward(ctx, g):
        from . import spectra
        target_audio, audio, *saved = ctx.saved_tensors
        fft_sizes, mag_weight, logmag_weight, l2, have = ctx.cfg
        cplx_list = saved[2 * len(fft_sizes):]
        cplx_iter = iter(cplx_list)
        cplx = [next(cplx_iter) if h else None for h in have]
        need_t, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g = g.float().reshape(-1).contiguous()   # one element, or one per clip
        grad_t = grad_v = None   # the first scale's kernel writes the gradient, the later ones add to it
        for i, size in enumerate(fft_sizes):
            hop = int(size * (1.0 - 0.75))
            win = spectra._cached_window(None, size, audio.device)
            t, v = saved[2 * i], saved[2 * i + 1]
            gt, gv = nat.spec_distance_backward(t, v, mag_weight, logmag_weight, g, 1.0, 1e-5, l2, need_target=need_t, need_value=need_v,
                                                per_row=ctx.per_item)
            if need_v:
                grad_v = nat.stft_mag_backward(audio, win, size, hop, gv, accumulate_into=grad_v, spec=cplx[i])
            if need_t:
                grad_t = nat.stft_mag_backward(target_audio, win, size, hop, gt, accumulate_into=grad_t)
        return grad_t, grad_v, None, None, None, None, None


MSS_FUSED = True   # module switch (tests compare both forms): MSSLoss on the GPU as TWO launches (sot_mss_loss_and_grad) instead of 36


class _MultiScaleSpectralFused(torch.autograd.Function):
    """MSSLoss as one call (include/sot_hip.h: sot_mss_loss_and_grad, csrc/sot_mss.hip): the loss of all scales AND its gradient w.r.t.
    the estimate's audio come out of the forward (no spectrogram is ever stored); the backward multiplies by the upstream gradient.
    Only the estimate is differentiated (trainer.py:206-221: the target is data); a target that asks for a gradient takes
    _MultiScaleSpectral."""

    @staticmethod
    def forward(ctx, target_audio, audio, fft_sizes, mag_weight, logmag_weight, l2, per_item=False):
        from . import spectra
        windows = [spectra._cached_window(None, size, audio.device) for size in fft_sizes]   # window=None -> hann (features.py:203-204)
        want = ctx.needs_input_grad[1]
        loss, grad = nat.mss_loss_and_grad(target_audio, audio, fft_sizes, windows, mag_weight, logmag_weight, 1e-5, l2, per_item, want)
        ctx.per_item = per_item
        if want:
            ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[1]:
            return (None,) * 7
        (grad,) = ctx.saved_tensors
        g = g.float()
        return None, grad * (g.reshape(-1, 1) if ctx.per_item else g), None, None, None, None, None


class MSSLoss(torch.nn.Module):
    """Multi-scale spectrogram loss, the reference's `losses.MSSLoss` (losses.py:365-425): for each FFT
    size the magnitude STFT (hann window, 75 % overlap, end-padded, normalized: features.compute_mag) of target and estimate,
    `mag_weight * mean D(t - v) + logmag_weight * mean D(safe_log t - safe_log v)` with D = |.| ('L1') or (.)^2 ('L2'),
    summed over the sizes.  On the GPU everything runs in HIP kernels behind one autograd node (also with `dims` = the two spectrogram
    axes: one value per clip); any other `dims`, FFT sizes the kernels do not cover and CPU tensors take the same composition on torch ops."""

    def __init__(self, fft_sizes=(2048, 1024, 512, 256, 128, 64), loss_type="L1", mag_weight=0.0, logmag_weight=0.0):
        super().__init__()
        self.fft_sizes = tuple(fft_sizes)
        self.loss_type = loss_type
        self.mag_weight = mag_weight
        self.logmag_weight = logmag_weight

    def forward(self, target_audio, audio, **kwargs):
        from . import spectra
        kind = self.loss_type.upper()
        if kind not in ("L1", "L2"):
            raise ValueError("Loss type ({}), must be " '"L1", "L2" '.format(kind))   # losses.py:36
        dims = kwargs.get("dims", None)
        # `dims` = the two spectrogram axes of [batch, freq, frames] (one value per clip): the per-row distance kernels.  (Any other
        # subset keeps an axis whose length differs from scale to scale -- with more than one FFT size the reference's `loss +=` fails.)
        per_item = (dims is not None and not isinstance(dims, int) and all(isinstance(d, int) and -3 <= d < 3 for d in dims)
                    and sorted(d % 3 for d in dims) == [1, 2])
        native_ok = (audio.is_cuda and target_audio.is_cuda and (dims is None or per_item) and audio.ndim == 2 and audio.shape == target_audio.shape and
                     audio.shape[0] > 0 and (self.mag_weight > 0 or self.logmag_weight > 0) and   # (an empty batch: torch's mean of nothing, NaN)
                     all(spectra.hip_stft_supported(s, int(s * 0.25), audio.shape[1]) for s in self.fft_sizes))
        if audio.is_cuda and not native_ok:
            warn_once(("mss", dims is not None, self.fft_sizes),
                      "MSSLoss: this call runs the torch composition instead of the HIP kernels (`dims` other than the two spectrogram axes, FFT sizes outside "
                      "64..4096, shapes that differ, or both weights zero)")
        if native_ok and MSS_FUSED and len(self.fft_sizes) <= 8 and all(int(s) in nat.MSS_FUSED_SIZES for s in self.fft_sizes) and \
                not (torch.is_grad_enabled() and target_audio.requires_grad):
            sizes = tuple(int(s) for s in self.fft_sizes)
            # rows the kernels can address: an expanded target (row stride 0) or an overlapping view is copied, as the reference's ops would
            target_audio, audio = nat.rows_view(target_audio), nat.rows_view(audio)
            glue = nat.glue() if (audio.dtype == torch.float32 and target_audio.dtype == torch.float32 and audio.stride(1) == 1 and
                                  target_audio.stride(1) == 1 and audio.shape[0] > 0) else None
            if glue is not None:   # ONE C++ call and a C++ autograd node (csrc/sot_torch_glue.cpp: MssLoss)
                wins = spectra._cached_windows(None, sizes, audio.device)
                return glue.mss_loss(target_audio, audio, wins, list(sizes), float(self.mag_weight), float(self.logmag_weight), kind == "L2", per_item)
            return _MultiScaleSpectralFused.apply(target_audio.float(), audio.float(), sizes, float(self.mag_weight),
                                                  float(self.logmag_weight), kind == "L2", per_item)
        if native_ok:
            return _MultiScaleSpectral.apply(target_audio.float(), audio.float(), self.fft_sizes, float(self.mag_weight),
                                             float(self.logmag_weight), kind == "L2", per_item)
        loss = 0.0
        for size in self.fft_sizes:
            hop = int(size * (1.0 - 0.75))
            t = spectra.stft_magnitude(target_audio, size, hop, None).permute(0, 2, 1)   # [batch, freq, frames] as compute_mag
            v = spectra.stft_magnitude(audio, size, hop, None).permute(0, 2, 1)
            for weight, a, b in ((self.mag_weight, t, v), (self.logmag_weight, _safe_log(t), _safe_log(v))):
                if weight > 0:
                    diff = a - b
                    red = list(range(diff.ndim)) if dims is None else dims
                    loss = loss + weight * (torch.mean(torch.abs(diff), dim=red) if kind == "L1" else torch.mean(diff ** 2, dim=red))
        return loss


def _safe_log(x, eps=1e-5):
    """utils.py:145-151."""
    e = torch.tensor(eps, device=x.device)
    return torch.log(torch.where(x <= e, e, x))


class Wasserstein1DWithTransform(torch.nn.Module):
    """The reference's audio-in form (losses.py:316-343): both signals go through the transform that
    ``features.get_transform(transform_kwargs, sample_rate)`` names, the transform's bin frequencies scaled to [0, 1]
    become the positions of both sides, and ``Wasserstein1D`` (attribute ``wasserstein``) compares the spectra.
    Built here for the STFT transform (``{"type": "stft", "n_fft", "hop_length", "window", "log", "sr"}``, the reference's
    ``features.TorchSTFT``, features.py:85-113: hann window unless a scipy window name is given; defaults n_fft 1024, hop
    256, sr 16000).  ``"cqt"`` (nnAudio) and ``"identity"`` (no bin frequencies in the reference either) raise.
    On the GPU the STFT, the loss and their backward are the HIP kernels of spectra.training_step_slice."""

    def __init__(self, p=1, fixed_x=None, require_sort=True, log_scaled_x=False, transform_kwargs=None, **kwargs):
        super().__init__()
        self.wasserstein = Wasserstein1D(p=p, fixed_x=fixed_x, require_sort=require_sort, log_scaled_x=log_scaled_x, **kwargs)
        if not isinstance(transform_kwargs, dict):
            raise AttributeError("transform_kwargs must be a dict (the reference pops 'sr' from it)")
        tk = dict(transform_kwargs)
        self.sr = tk.pop("sr", 16000)
        name = tk.pop("type")
        if name != "stft":
            raise ValueError(f"Unknown transform {name}" if name not in ("cqt", "identity") else
                             f"transform {name!r} is not built here (only 'stft')")
        self.n_fft = int(tk.pop("n_fft", 1024))
        hop_length = tk.pop("hop_length", 256)
        self.log = bool(tk.pop("log", False))
        self.window = tk.pop("window", None)
        overlap = 1 - hop_length / self.n_fft                 # features.py:99, then features.py:194
        self.hop = int(self.n_fft * (1.0 - overlap))
        assert self.n_fft * overlap % 2.0 == 0.0              # features.py:199

    def forward(self, x, y, **kwargs):
        from . import spectra
        if not self.log and not kwargs:
            return spectra.training_step_slice(self.wasserstein, x, y, n_fft=self.n_fft, hop=self.hop, sample_rate=float(self.sr),
                                               window=self.window)
        spec_x = spectra.stft_magnitude(x, self.n_fft, self.hop, self.window)
        spec_y = spectra.stft_magnitude(y, self.n_fft, self.hop, self.window)
        if self.log:
            spec_x, spec_y = _safe_log(spec_x), _safe_log(spec_y)
        pos = spectra.unit_frequencies(self.n_fft, float(self.sr), x.device)
        return self.wasserstein(spec_x, spec_y, x_pos=pos, y_pos=pos.clone(), **kwargs)


class MixOfLosses(torch.nn.Module):
    """losses.py:346-362: weighted dict of losses keyed by class name."""

    def __init__(self, losses, weights=None):
        super().__init__()
        self.losses = losses
        self.weights = weights

    def forward(self, x, y, **kwargs):
        loss = {}
        for loss_fn, weight in zip(self.losses, self.weights):
            loss_ = loss_fn(x, y, **kwargs) * weight
            loss[loss_fn.__class__.__name__] = loss_
        return loss


# ---- the other operands a MixOfLosses configuration of the reference may name (losses.py:7-86).  Not on the OT path and not HIP: plain
# ---- torch-op compositions with the reference's call signature, so that a configuration that mixes them with Wasserstein1D loads.

def mean_difference(target, value, loss_type="L1", weights=None, dims=None):
    """losses.py:7-36: mean over `dims` (all axes when None) of |d·w| ("L1") or d²·w ("L2"), d = target − value."""
    kind = loss_type.upper()
    if kind not in ("L1", "L2"):
        raise ValueError('Loss type ({}), must be "L1", "L2" '.format(kind))
    delta = target - value
    w = 1.0 if weights is None else weights
    term = (delta * w).abs() if kind == "L1" else delta ** 2 * w
    return torch.mean(term, dim=list(range(term.ndim)) if dims is None else dims)


class MeanDifference(torch.nn.Module):
    """losses.py:39-54: mean_difference of the two inputs, optionally after sorting each along its last axis."""

    def __init__(self, loss_type="L1"):
        super().__init__()
        self.loss_type = loss_type

    def forward(self, x, y, weights=None, sort=False, **kwargs):
        if sort:
            x, y = torch.sort(x, dim=-1)[0], torch.sort(y, dim=-1)[0]
        return mean_difference(x, y, loss_type=self.loss_type, weights=weights, dims=kwargs.get("dims", None))


class KL(torch.nn.Module):
    """losses.py:57-86: rows normalised to unit mass (safe_divide), KL(input ‖ target) = Σ a·(log(a + eps) − log(b + eps)) per row
    (`reverse` swaps the roles), mean over `dims`."""

    def __init__(self, eps=1e-10, **kwargs):
        super().__init__()
        self.eps = eps
        self.reverse = kwargs.get("reverse", False)

    def forward(self, input, target, **kwargs):
        lead = input.shape[:-1]
        rows = lambda t: t.reshape(-1, t.shape[-1]) if t.ndim == 3 else t   # noqa: E731
        a, b = rows(input), rows(target)
        if self.reverse:
            a, b = b, a
        a = safe_divide(a, a.sum(dim=-1, keepdim=True))
        b = safe_divide(b, b.sum(dim=-1, keepdim=True))
        per_row = (a * (torch.log(a + self.eps) - torch.log(b + self.eps))).sum(dim=-1)
        return torch.mean(per_row.reshape(lead), dim=kwargs.get("dims", None))
