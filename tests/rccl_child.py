"""Helper for tests/test_rccl_gpu.py: ONE rank of a `python -m torch.distributed.run` job with backend "nccl" (RCCL on ROCm).
Evaluates sharded_sot_loss on this rank's row block (HIP kernels + the RCCL reductions of sot_amd.distributed) and the
single-process module on the whole batch, and prints one JSON line with both (rank 0)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from sot_amd import _native as nat
    from sot_amd.distributed import shard_rows, sharded_sot_loss
    from sot_amd.losses import Wasserstein1D
    nat.load(build_if_missing=False)
    g = torch.Generator().manual_seed(5)
    x, y = torch.rand(24, 4, 257, generator=g) ** 4, torch.rand(24, 4, 257, generator=g) ** 4
    f = torch.fft.rfftfreq(512, d=1.0 / 16000.0)
    pos = (f / f.max()).float().to(dev)
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
    a, b = shard_rows(x.shape[0], rank, world)
    out = {}
    for tag, dims in (("mean", None), ("dims1", [1])):
        yl = y[a:b].to(dev).requires_grad_(True)
        got = sharded_sot_loss(mod, x[a:b].to(dev), yl, x_pos=pos, y_pos=pos.clone(), dims=dims)
        w = torch.arange(1, got.numel() + 1, dtype=torch.float32, device=dev).reshape(got.shape)
        (got * w).sum().backward()
        yf = y.to(dev).requires_grad_(True)
        want = mod(x.to(dev), yf, x_pos=pos, y_pos=pos.clone(), dims=dims)
        (want * w).sum().backward()
        torch.cuda.synchronize()
        out[tag] = {"got": got.detach().reshape(-1).cpu().tolist(), "want": want.detach().reshape(-1).cpu().tolist(),
                    "grad_err": float((yl.grad - yf.grad[a:b]).abs().max()), "grad_max": float(yf.grad.abs().max())}
    # the raw collective on a device tensor (what bench.py's N > 1 step issues)
    t = torch.full((1,), float(rank + 1), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out["allreduce"] = float(t)
    out["backend"] = dist.get_backend()
    out["world"] = world
    dist.barrier()
    if rank == 0:
        print("RCCL_CHILD " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
