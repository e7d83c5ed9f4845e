import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29557")
import torch, torch.distributed as dist
from sot_amd import _native as nat
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
nat.load(build_if_missing=False)
B, N = 8192, 2048
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(4)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
rows = torch.empty(B, device=dev); ring = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(4)]
lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(n, two, coll):
    acc = {"fwd": 0.0, "mean": 0.0, "ar": 0.0}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if two: torch.cuda.set_stream(lanes[i & 1])
        a = time.perf_counter(); nat.forward_rows(*sets[i % 4], pos, pos2, 1.0, 8, plan, rows)
        b = time.perf_counter(); nat.reduce_mean(rows, sum_out=ring[i % 4])
        c = time.perf_counter()
        if coll: dist.all_reduce(ring[i % 4], op=dist.ReduceOp.SUM)
        d = time.perf_counter()
        acc["fwd"] += b - a; acc["mean"] += c - b; acc["ar"] += d - c
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    torch.cuda.set_stream(torch.cuda.default_stream())
    return {k: round(v / n * 1e6, 1) for k, v in acc.items()}, round(host / n * 1e6, 1), round(tot / n * 1e6, 1)
for two in (False, True):
    for coll in (False, True):
        run(200, two, coll)
        print("two streams" if two else "one stream ", "all_reduce" if coll else "no collective", run(500, two, coll))
dist.destroy_process_group()
