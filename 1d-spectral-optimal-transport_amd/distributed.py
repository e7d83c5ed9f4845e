"""Row-sharded SOT loss across the GPUs of one node (one process per GPU, RCCL over xGMI).

Rows are independent (losses.py:273-313 has no cross-row op before the final mean at :211), so the
batch shards into contiguous row blocks with NO data-path collective; the only exchange is one
all-reduce(SUM) of a single fp64 scalar -- each rank's sum of row losses -- after which every rank
divides by the global row count (reproducing ``torch.mean`` with dims=None).  Input gradients need no
collective: each rank scales its local gradients by 1/B_global.

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


def sharded_sot_loss(loss_module, x_local, y_local, x_pos=None, y_pos=None, group=None, **kwargs):
    """Global-batch mean of the SOT loss when each rank holds a contiguous block of rows.

    `loss_module` is a `Wasserstein1D`; `x_local`/`y_local` are this rank's rows.  Equivalent to calling
    the module on the concatenated batch (dims=None): local kernel -> local fp64 sum -> all-reduce of
    ONE scalar -> divide by the global row count."""
    rows = loss_module.row_losses(x_local, y_local, x_pos=x_pos, y_pos=y_pos, **kwargs)
    local_sum = _RowSum.apply(rows)  # differentiable; bench.py uses the fused no-grad form (nat.loss_fused)
    return global_mean_from_local_sum(local_sum, rows.numel(), group)
