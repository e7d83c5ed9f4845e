"""CPU, world_size 2, gloo: the row-sharded reduction logic of sot_amd.distributed (the N>1 path).
Row losses come from the oracle (checker) so that no GPU is needed; on GPUs the same function is fed
by the HIP kernel and `backend='nccl'` (RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, rows_np, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sot_amd.distributed import global_mean_from_local_sum, shard_rows
    a, b = shard_rows(len(rows_np), rank, world)
    local = torch.tensor(rows_np[a:b], dtype=torch.float32, requires_grad=True)
    local_sum = local.double().sum()
    mean = global_mean_from_local_sum(local_sum, b - a)
    mean.backward()
    q.put((rank, float(mean), local.grad.numpy().copy(), (a, b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 65])
def test_sharded_mean_equals_global_mean_world2(total):
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    x, y = gen_inputs("peaky", total, 96, 96, 4)
    pos = np.linspace(0, 1, 96, dtype=np.float32)
    rows = so.forward(x.numpy(), y.numpy(), pos, pos, p=1.0, flags=so.make_flags())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = float(np.mean(rows.astype(np.float64)))
    for rank, mean, grad, (a, b) in out:
        assert abs(mean - want) <= 1e-6 * abs(want)
        np.testing.assert_allclose(grad, np.full(b - a, 1.0 / total, np.float32), rtol=1e-6)
