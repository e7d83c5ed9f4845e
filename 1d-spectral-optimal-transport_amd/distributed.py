"""Row-sharded SOT loss across the GPUs of one node (one process per GPU, RCCL over xGMI).

Rows are independent (losses.py:273-313 has no cross-row op before the final mean at :211), so the
batch shards into contiguous row blocks with NO data-path collective; the only exchange is one
all-reduce(SUM) of a single fp64 scalar -- each rank's sum of row losses -- after which every rank
divides by the global row count (reproducing ``torch.mean`` with dims=None), or, for a ``dims``
reduction (losses.py:208-211), one all-gather of the per-row losses.  Input gradients need no
collective: each rank scales its local gradients by 1/B_global.  Parameter gradients are per-rank
partial sums (SUM-reduce them; see ``global_loss_from_local_rows`` for DistributedDataParallel).

The reference is single-device (every YAML: devices: 1, strategy: null), so this is new
functionality, not a port.  `backend="nccl"` is RCCL on ROCm; the same code runs on `gloo` (CPU
tests exercise the sharding/reduction logic with world_size 2).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(total_rows: int, rank: int, world_size: int):
    """Contiguous row block [start, stop) owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(total_rows, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class _AllReduceSumCount(torch.autograd.Function):
    """(sum, count) -> global mean with ONE all-reduce of a 2-element fp64 tensor.
    d(mean)/d(local_sum) = 1/global_count on every rank (each rank differentiates its own rows)."""

    @staticmethod
    def forward(ctx, local_sum, local_rows, group):
        packed = torch.stack([local_sum.detach().to(torch.float64).reshape(()),
                              torch.tensor(float(local_rows), dtype=torch.float64, device=local_sum.device)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        ctx.save_for_backward(packed)
        return packed[0] / packed[1]

    @staticmethod
    def backward(ctx, g):
        (packed,) = ctx.saved_tensors
        return g / packed[1], None, None


def global_mean_from_local_sum(local_sum: torch.Tensor, local_rows: int, group=None) -> torch.Tensor:
    """Global mean from per-rank (sum of row losses, row count); differentiable w.r.t. local_sum.
    Returns a float32 0-d tensor (the dtype of torch.mean of fp32 row losses)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return (local_sum / local_rows).to(torch.float32)
    return _AllReduceSumCount.apply(local_sum, local_rows, group).to(torch.float32)


class _RowSum(torch.autograd.Function):
    """fp64 fixed-order sum of the local row losses on the GPU (sot_w1d_reduce_mean's sum_out)."""

    @staticmethod
    def forward(ctx, rows):
        from . import _native as nat
        ctx.count = rows.numel()
        _, total = nat.reduce_mean(rows, want_sum=True)
        return total

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.float32).expand(ctx.count)


class _GatherRows(torch.autograd.Function):
    """All-gather of per-rank loss blocks along dim 0 (blocks may differ in length).  Every rank ends up with the global
    tensor; the backward hands each rank the slice of the upstream gradient that belongs to its own block (no collective:
    every rank evaluates the same global reduction)."""

    @staticmethod
    def forward(ctx, local, group):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
        counts = [int(c) for c in counts]
        longest = max(counts)
        padded = local.detach()
        if local.shape[0] < longest:
            padded = torch.cat([padded, padded.new_zeros((longest - local.shape[0],) + tuple(local.shape[1:]))])
        blocks = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(blocks, padded.contiguous(), group=group)
        ctx.start, ctx.count = sum(counts[:rank]), counts[rank]
        return torch.cat([b[:c] for b, c in zip(blocks, counts)])

    @staticmethod
    def backward(ctx, g):
        return g[ctx.start:ctx.start + ctx.count], None


def global_loss_from_local_rows(local_loss: torch.Tensor, dims=None, group=None, ddp_average: bool = False) -> torch.Tensor:
    """The reduction of losses.py:207-211 over a batch whose dim 0 is sharded across ranks.

    `local_loss`: this rank's block of per-row losses, shaped like the leading dims of its inputs ([b_local] or
    [b_local, time]).  dims=None (or []): the mean over every row of the global batch -- ONE all-reduce of the fp64 (sum,
    count) pair.  Any other `dims`: the blocks are all-gathered along dim 0 and `torch.mean(global, dim=dims)`
    is evaluated on every rank, exactly what the reference computes on the concatenated batch.

    Gradients: each rank receives d(global loss)/d(its rows), with no collective.  Gradients of shared PARAMETERS are
    therefore per-rank partial sums: SUM-reduce them across ranks to get the single-process gradient.  A wrapper that
    AVERAGES parameter gradients instead (torch DistributedDataParallel) yields them world_size times too small; pass
    ddp_average=True to scale this loss's backward by world_size so that the average is the single-process gradient."""
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if on else 1
    if dims is None or (isinstance(dims, (list, tuple)) and len(dims) == 0):
        flat = local_loss.reshape(-1)
        if flat.is_cuda:
            local_sum = _RowSum.apply(flat.float().contiguous())   # fp64 fixed-order sum on the GPU (sot_w1d_reduce_mean)
        else:
            local_sum = flat.double().sum()
        out = global_mean_from_local_sum(local_sum, flat.numel(), group)
    else:
        glob = _GatherRows.apply(local_loss, group) if on else local_loss
        out = torch.mean(glob, dim=dims)
    if ddp_average and world > 1:
        out = _ScaleGrad.apply(out, float(world))
    return out


class _ScaleGrad(torch.autograd.Function):
    """identity in the forward, gradient times `factor` in the backward"""

    @staticmethod
    def forward(ctx, value, factor):
        ctx.factor = factor
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.factor, None


def sharded_sot_loss(loss_module, x_local, y_local, x_pos=None, y_pos=None, group=None, ddp_average=False, **kwargs):
    """SOT loss of the GLOBAL batch when each rank holds a contiguous block of rows (of dim 0) of it.

    `loss_module` is a `Wasserstein1D`; `x_local`/`y_local` are this rank's rows ([b_local, N] or [b_local, time, N]).
    Equivalent to calling the module on the concatenated batch: local kernel -> per-row losses -> the reduction of
    `global_loss_from_local_rows` (dims=None: one all-reduce of a scalar pair; `dims=...`: all-gather of the row blocks).
    See there for how parameter gradients must be reduced (`ddp_average`)."""
    dims = kwargs.pop("dims", None)
    rows = loss_module.row_losses(x_local, y_local, x_pos=x_pos, y_pos=y_pos, **kwargs)   # flat [local rows], hinge applied
    return global_loss_from_local_rows(rows.reshape(x_local.shape[:-1]), dims, group, ddp_average)
