"""CPU, world_size 2, gloo: the row-sharded reduction logic of sot_amd.distributed (the N>1 path).
Row losses come from the oracle (checker) so that no GPU is needed; on GPUs the same functions are fed
by the HIP kernel and `backend='nccl'` (RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker(rank, world, port, rows_np, q):
    _init(rank, world, port)
    from sot_amd.distributed import global_mean_from_local_sum, shard_rows
    a, b = shard_rows(len(rows_np), rank, world)
    local = torch.tensor(rows_np[a:b], dtype=torch.float32, requires_grad=True)
    local_sum = local.double().sum()
    mean = global_mean_from_local_sum(local_sum, b - a)
    mean.backward()
    q.put((rank, float(mean), local.grad.numpy().copy(), (a, b)))
    dist.barrier()
    dist.destroy_process_group()


def _run(target, args, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (abs(hash(target.__name__)) + sum(map(ord, repr(args[:1])))) % 400
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def _oracle_rows(total, n=96, seed=4):
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    x, y = gen_inputs("peaky", total, n, n, seed)
    pos = np.linspace(0, 1, n, dtype=np.float32)
    return so.forward(x.numpy(), y.numpy(), pos, pos, p=1.0, flags=so.make_flags())


@pytest.mark.parametrize("total", [64, 65])
def test_sharded_mean_equals_global_mean_world2(total):
    rows = _oracle_rows(total)
    out = _run(_worker, (rows,))
    want = float(np.mean(rows.astype(np.float64)))
    for rank, mean, grad, (a, b) in out:
        assert abs(mean - want) <= 1e-6 * abs(want)
        np.testing.assert_allclose(grad, np.full(b - a, 1.0 / total, np.float32), rtol=1e-6)


def _dims_worker(rank, world, port, rows_np, dims, q):
    """rows_np: [batch, time] global per-row losses; dim 0 is sharded (65 rows: uneven blocks)."""
    _init(rank, world, port)
    from sot_amd.distributed import global_loss_from_local_rows, shard_rows
    a, b = shard_rows(rows_np.shape[0], rank, world)
    local = torch.tensor(rows_np[a:b], dtype=torch.float32, requires_grad=True)
    out = global_loss_from_local_rows(local, dims=dims)
    weights = torch.arange(1, out.numel() + 1, dtype=torch.float32).reshape(out.shape)   # a non-trivial upstream gradient
    (out * weights).sum().backward()
    q.put((rank, out.detach().numpy().copy(), local.grad.numpy().copy(), (a, b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dims", [[0], [1], [0, 1], None, []])
def test_dims_reduction_all_gathers_the_row_blocks_world2(dims):
    """losses.py:208-211 with `dims`: every rank evaluates torch.mean(global_loss, dim=dims) on the all-gathered blocks
     and receives the gradient slice of its own rows; dims None / [] is the scalar all-reduce path."""
    rows = _oracle_rows(65 * 3).reshape(65, 3)
    out = _run(_dims_worker, (rows, dims))
    g = torch.tensor(rows, requires_grad=True)
    want = torch.mean(g, dim=dims) if dims else torch.mean(g)
    weights = torch.arange(1, want.numel() + 1, dtype=torch.float32).reshape(want.shape)
    (want * weights).sum().backward()
    for rank, got, grad, (a, b) in out:
        np.testing.assert_allclose(got, want.detach().numpy(), rtol=2e-6)
        np.testing.assert_allclose(grad, g.grad.numpy()[a:b], rtol=2e-6)


def _ddp_worker(rank, world, port, feats_np, ddp_average, q):
    """A tiny model under DistributedDataParallel (which AVERAGES parameter gradients): per-row 'losses' are a
    differentiable function of a shared parameter; the global mean goes through global_loss_from_local_rows."""
    _init(rank, world, port)
    from sot_amd.distributed import global_loss_from_local_rows, shard_rows
    torch.manual_seed(0)
    model = torch.nn.Linear(feats_np.shape[1], 1, bias=False)
    with torch.no_grad():
        model.weight.copy_(torch.linspace(-1, 1, feats_np.shape[1]).reshape(1, -1))
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    a, b = shard_rows(feats_np.shape[0], rank, world)
    rows = ddp(torch.tensor(feats_np[a:b])).pow(2).reshape(-1)
    global_loss_from_local_rows(rows, ddp_average=ddp_average).backward()
    q.put((rank, model.weight.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_parameter_gradients_under_ddp_need_ddp_average():
    feats = np.random.RandomState(0).randn(64, 5).astype(np.float32)
    w = torch.linspace(-1, 1, 5).reshape(1, -1).requires_grad_(True)
    (torch.tensor(feats) @ w.t()).pow(2).mean().backward()
    single = w.grad.numpy()
    for rank, grad in _run(_ddp_worker, (feats, True)):
        np.testing.assert_allclose(grad, single, rtol=1e-5)            # averaged by DDP, scaled back by ddp_average
    for rank, grad in _run(_ddp_worker, (feats, False)):
        np.testing.assert_allclose(grad * 2, single, rtol=1e-5)        # without it: world_size times too small


def _module_worker(rank, world, port, x_np, y_np, dims, q):
    """sharded_sot_loss end to end on CPU tensors: module -> per-row losses (the package's torch-op route) -> reduction."""
    _init(rank, world, port)
    from sot_amd.distributed import shard_rows, sharded_sot_loss
    from sot_amd.losses import Wasserstein1D
    a, b = shard_rows(x_np.shape[0], rank, world)
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
    pos = torch.linspace(0, 1, x_np.shape[-1])
    yl = torch.tensor(y_np[a:b], requires_grad=True)
    out = sharded_sot_loss(mod, torch.tensor(x_np[a:b]), yl, x_pos=pos, y_pos=pos.clone(), dims=dims)
    w = torch.arange(1, out.numel() + 1, dtype=torch.float32).reshape(out.shape)
    (out * w).sum().backward()
    q.put((rank, out.detach().numpy().copy(), yl.grad.numpy().copy(), (a, b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dims", [None, [1]])
def test_sharded_module_equals_single_process_module_world2(dims):
    """The N > 1 path from the module down: two ranks with uneven row blocks of a [5, 3, 64] batch give the loss and
    the input gradients of one process on the whole batch."""
    from sot_amd.losses import Wasserstein1D
    g = torch.Generator().manual_seed(11)
    x, y = torch.rand(5, 3, 64, generator=g) ** 4, torch.rand(5, 3, 64, generator=g) ** 4
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
    pos = torch.linspace(0, 1, 64)
    yf = y.clone().requires_grad_(True)
    want = mod(x, yf, x_pos=pos, y_pos=pos.clone(), dims=dims)
    w = torch.arange(1, want.numel() + 1, dtype=torch.float32).reshape(want.shape)
    (want * w).sum().backward()
    for rank, got, grad, (a, b) in _run(_module_worker, (x.numpy(), y.numpy(), dims)):
        np.testing.assert_allclose(got, want.detach().numpy(), rtol=2e-6)
        np.testing.assert_allclose(grad, yf.grad.numpy()[a:b], rtol=2e-5, atol=1e-9)
