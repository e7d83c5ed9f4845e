"""Which distributed step structure is cheapest on the host?  Run under torch.distributed.run (1+ ranks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from sot_amd import _native as nat
from sot_amd.losses import Wasserstein1D
lr = int(os.environ.get("LOCAL_RANK", "0")); torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
dev = torch.device("cuda", lr); world = dist.get_world_size()
B, N = 8192, 2048
g = torch.Generator(device=dev).manual_seed(lr)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(6)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
mod = Wasserstein1D(p=1).to(dev)
x2, y2, xp, yp, flags, plan, _ = mod._marshal(sets[0][0], sets[0][1], pos, pos2, {})
ring = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(4)]
inv = 1.0 / (world * B)

def local(i):
    x, y = sets[i % 6]
    rows = nat.forward_rows(x, y, xp, yp, 1.0, flags, plan)
    buf = ring[i % 4]
    nat.reduce_mean(rows, sum_out=buf)
    return buf

def sync_step(i):
    buf = local(i); dist.all_reduce(buf); return buf * inv

pend = [None]
def async_step(i):
    buf = local(i); w = dist.all_reduce(buf, async_op=True); prev = pend[0]; pend[0] = (w, buf)
    if prev is not None:
        prev[0].wait(); return prev[1] * inv

comm = torch.cuda.Stream(device=dev)
done = [None] * 4
def side_step(i):
    s = i % 4
    if done[s] is not None:
        torch.cuda.current_stream().wait_event(done[s])   # the slot's previous all-reduce has finished
    buf = local(i)
    ev = torch.cuda.Event(); ev.record()
    comm.wait_event(ev)
    with torch.cuda.stream(comm):
        dist.all_reduce(buf)
        d = torch.cuda.Event(); d.record(comm)
    done[s] = d
    return buf

ready2 = [torch.cuda.Event() for _ in range(4)]
done2 = [torch.cuda.Event() for _ in range(4)]
used2 = [False] * 4
rowbuf = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(4)]
main_stream = torch.cuda.current_stream()
pg = dist.group.WORLD
def lean_step(i):
    s = i % 4
    if used2[s]:
        main_stream.wait_event(done2[s])
    x, y = sets[i % 6]
    rows = nat.forward_rows(x, y, xp, yp, 1.0, flags, plan, rowbuf[s])
    buf = ring[s]
    nat.reduce_mean(rows, sum_out=buf)
    ready2[s].record(main_stream)
    comm.wait_event(ready2[s])
    torch.cuda.set_stream(comm)
    dist.all_reduce(buf)
    torch.cuda.set_stream(main_stream)
    done2[s].record(comm)
    used2[s] = True
    return buf

alt = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
def alt_step(i):
    s = i % 4
    torch.cuda.set_stream(alt[i & 1])
    x, y = sets[i % 6]
    rows = nat.forward_rows(x, y, xp, yp, 1.0, flags, plan, rowbuf[s])
    buf = ring[s]
    nat.reduce_mean(rows, sum_out=buf)
    dist.all_reduce(buf)
    return buf

def timeit(name, fn, n=200, post=None):
    for i in range(20): fn(i)
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    host = (time.perf_counter() - t0) / n
    if post: post()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    if dist.get_rank() == 0: print(f"{name:40s} {dt * 1e6:7.1f} us/step   (host enqueue {host * 1e6:6.1f} us/step)")

with torch.no_grad():
    timeit("local kernels only (no collective)", local)
    timeit("sync all_reduce", sync_step)
    timeit("async all_reduce, 1-step pipeline", async_step, post=lambda: (pend[0][0].wait(), pend.__setitem__(0, None)))
    timeit("sync all_reduce on a side stream (ring of 4)", side_step, post=lambda: torch.cuda.current_stream().wait_stream(comm))
    timeit("lean side stream (set_stream, reused events/buffers)", lean_step, post=lambda: torch.cuda.current_stream().wait_stream(comm))
    main_stream.wait_stream(comm)
    for st in alt: st.wait_stream(main_stream)
    timeit("two alternating compute streams, sync all_reduce", alt_step, post=lambda: torch.cuda.set_stream(main_stream))
    torch.cuda.synchronize()
    # whole step in a HIP graph (static input set): forward + reduce + all_reduce + scale
    try:
        sx, sy = sets[0]
        gbuf = torch.zeros(1, dtype=torch.float64, device=dev)
        def gstep():
            rows = nat.forward_rows(sx, sy, xp, yp, 1.0, flags, plan); nat.reduce_mean(rows, sum_out=gbuf); dist.all_reduce(gbuf); return gbuf * inv
        gstep(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = gstep()
        timeit("HIP graph replay of the whole step", lambda i: gr.replay())
    except Exception as e:
        if dist.get_rank() == 0: print("graph capture of the collective failed:", repr(e)[:200])
dist.barrier(); dist.destroy_process_group()
