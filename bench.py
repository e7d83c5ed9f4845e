#!/usr/bin/env python3
"""bench.py -- SOT-loss evaluations/s on [B, N_fft] spectrum pairs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic spectrum pairs already resident in
HBM: Wasserstein1D forward (fused HIP kernel + fixed-order mean).  With N>1 every rank owns its own
B rows (weak scaling, no data-path collective) and the step ends with ONE RCCL all-reduce of the
scalar partial sum.  Inputs rotate over several distinct sets so that the 256 MiB Infinity Cache
cannot serve them.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- dominant kernel (sot_forward_full_kernel: the forward specialised for rows that fill their geometry): algorithmic bytes per launch / average launch
                  duration measured with HIP events on the launch stream inside the timed region;
  cpu_baseline -- the op-for-op torch restatement of the reference (oracle/torch_restatement.py,
                  kind "port") timed on this host's cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODES = {
    "p1": dict(p=1),  # north_star: L1-normalised Wasserstein-1
    "cutoff": dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True),  # paper SOT-2048
    "nocut": dict(p=2, square_dist=True),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md "Chip-level parameters": HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm", type=int, default=400,
                    help="untimed steps before the W warm-up steps: the first few hundred launches run ~10 %% slower (clock ramp)")
    ap.add_argument("--rows", type=int, default=8192, help="rows (spectrum pairs) per GPU")
    ap.add_argument("--nfft", type=int, default=2048, help="row length N")
    ap.add_argument("--mode", choices=list(MODES), default="p1")
    ap.add_argument("--sets", type=int, default=6, help="distinct rotating input sets (>= 4 defeats the 256 MiB L3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph-steps", action="store_true", help="replay each step from a HIP graph (opt-in)")
    ap.add_argument("--lanes", type=int, default=0, choices=[0, 1, 2],
                    help="HIP streams the steps alternate over: 1 = one stream; 2 = the mean kernel / all-reduce of a step "
                         "overlaps the next step's forward kernel; 0 (default) = 1 on one GPU (per-kernel times then agree "
                         "with rocprofv3), 2 when there is a collective to hide (N > 1)")
    ap.add_argument("--cpu-rows", type=int, default=2048)
    return ap.parse_args()


def cpu_baseline(mode, n, rows, seed):
    """The reference's CPU PyTorch path (op-for-op restatement) on this host's cores, bounded sample.
    ATen's intra-op scaling on this problem saturates well below a big host's core count, so a few thread
    counts are tried (within a ~25 s budget) and the best is reported with the count actually used."""
    from oracle import torch_restatement as tr
    from oracle.inputs import gen_inputs
    ncpu = os.cpu_count() or 1
    x, y = gen_inputs("uniform", rows, n, n, seed)
    pos = torch.linspace(0, 1, n)
    c = MODES[mode]
    kw = dict(p=c.get("p", 1), square_dist=c.get("square_dist", False), dont_normalize=c.get("dont_normalize", False),
              limit_quantile_range=c.get("limit_quantile_range", False))
    best, best_threads, val, tried = float("inf"), 1, 0.0, []
    t_end = time.perf_counter() + 25.0
    for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64, ncpu)}):
        if time.perf_counter() > t_end:
            break
        torch.set_num_threads(threads)
        with torch.no_grad():
            tr.sot_loss(x, y, pos, pos.clone(), **kw)  # warm-up
            for _ in range(2):
                t0 = time.perf_counter()
                val = tr.sot_loss(x, y, pos, pos.clone(), **kw)
                dt = time.perf_counter() - t0
                if dt < best:
                    best, best_threads = dt, threads
        tried.append(threads)
    return {"value": rows / best, "unit": "rows/s", "cores": best_threads, "kind": "port",
            "sample": f"{rows} rows x N={n}, mode {mode}: best call of oracle/torch_restatement.sot_loss over thread counts "
                      f"{tried} on a {ncpu}-cpu host (torch {torch.__version__} CPU)", "scalar": float(val)}


# Launches of the timed region that are bracketed by HIP events: every 4th with one stream (an event pair costs ~2 us of
# stream time), every 16th with two alternating streams (the bracketed launch is isolated from the other stream: a pipeline
# bubble of ~12 us; per-step time 52.4 us at stride 4, 43.2 us at stride 16).
def event_stride(n_lanes):
    return int(os.environ.get("SOT_BENCH_EVENT_STRIDE", "4" if n_lanes == 1 else "16"))


def main():
    args = parse()
    # stdout carries exactly ONE line (rank 0's JSON): native libraries print there too (RCCL's version banner at
    # communicator creation), so file descriptor 1 is pointed at stderr for the run and the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SOT_BENCH_FORCE_DIST=1 exercises the RCCL branch even with one rank (used to test it on a 1-GPU box)
    dist_on = world > 1 or (os.environ.get("SOT_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev = torch.device("cuda", local_rank if dist_on else 0)
    torch.cuda.set_device(dev)

    from sot_amd import _native as nat
    from sot_amd.losses import Wasserstein1D
    from oracle.inputs import gen_inputs

    nat.load(build_if_missing=False)  # must be the prebuilt in-tree library
    B, N = args.rows, args.nfft
    mod = Wasserstein1D(**MODES[args.mode]).to(dev)
    pos_x = torch.linspace(0, 1, N, device=dev)
    pos_y = pos_x.clone()

    # set 0: the BASELINE recipe (CPU generator, seed 1234 + rank, x drawn before y); others: device RNG
    x0, y0 = gen_inputs("uniform", B, N, N, 1234 + rank)
    sets = [(x0.to(dev), y0.to(dev))]
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    for _ in range(args.sets - 1):
        sets.append((torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)))

    # The rotating inputs are fixed tensors: their marshalled form (contiguous [B, N] views, flag word, position plan) is
    # computed once; a step then is exactly the FFI calls Wasserstein1D.forward makes (no GPU work is skipped).
    marshalled = [mod._marshal(x, y, pos_x, pos_y, {}) for x, y in sets]

    # Steps are independent.  With N > 1 (or --lanes 2) they are issued on TWO alternating HIP streams: one step's batch-mean
    # kernel and its RCCL all-reduce then overlap the next step's forward kernel instead of leaving the GPU idle for the
    # collective's latency.  Every 16th step of the timed region is then bracketed by HIP events on the stream it runs on and
    # isolated from the other stream (it waits for it, and the other stream's next step waits for the closing event), so the
    # event pair times the forward kernel alone.  Measured with one rank through RCCL: 61.7 us/step with everything on one
    # stream, 44-47 us like this.  All work, collectives included, completes inside the timed region (device-wide
    # synchronize at its end).  On one GPU the default is ONE stream: consecutive forward kernels then never share the GPU
    # and rocprofv3's per-kernel durations agree with the event-bracketed ones (--lanes 2 on one GPU: 42.9 instead of
    # 48.5 us per step, 191 instead of 169 M rows/s, but overlapped kernel durations in a profile).
    graph_mode = bool(args.graph_steps)   # --graph-steps: single stream, the whole step replayed from a HIP graph (opt-in)
    n_lanes = args.lanes if args.lanes else (2 if dist_on else 1)
    if graph_mode:
        n_lanes = 1
    EVENT_STRIDE = event_stride(n_lanes)
    default_stream = torch.cuda.current_stream()
    lanes = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    ring = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(4)]   # local / global fp64 sums (N > 1)
    rowbuf = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(4)]  # row losses; slot i % 4 stays on one lane
    inv_global_rows = 1.0 / float(world * B)   # weak scaling: every rank owns exactly B rows

    def step(i, profile=None):
        with torch.no_grad():
            x2, y2, xp, yp, flags, plan, _ = marshalled[i % len(sets)]
            slot = i % len(ring)
            if n_lanes == 1:
                cur = torch.cuda.current_stream()
            else:
                cur, other = lanes[i & 1], lanes[1 - (i & 1)]
                torch.cuda.set_stream(cur)
            if profile is not None:
                a, b = profile
                if n_lanes == 2:
                    cur.wait_stream(other)
                a.record(cur)
            # same kernels as Wasserstein1D.forward (sot_w1d_loss), issued as two calls so that the HIP events bracket the
            # dominant kernel (sot_forward_full_kernel: the forward specialised for rows that fill their geometry) alone
            rows = nat.forward_rows(x2, y2, xp, yp, float(mod.p), flags, plan, rowbuf[slot])
            if profile is not None:
                b.record(cur)
                if n_lanes == 2:
                    other.wait_event(b)
            if not dist_on:
                return nat.reduce_mean(rows)
            # N > 1: local kernels -> ONE all-reduce(SUM) of the fp64 partial sum over RCCL -> global mean
            import torch.distributed as dist
            nat.reduce_mean(rows, sum_out=ring[slot])
            dist.all_reduce(ring[slot], op=dist.ReduceOp.SUM)   # in place: the slot then holds the global sum
            return ring[slot]   # the mean is ring[slot] * inv_global_rows (applied where the value is read)

    def enter_lanes():
        for ln in lanes:
            ln.wait_stream(default_stream)

    def leave_lanes():
        torch.cuda.synchronize()
        torch.cuda.set_stream(default_stream)

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()

    enter_lanes()
    for i in range(args.prewarm + args.warmup):
        out = step(i)
    first_t = step(0)  # parity value on set 0 (global mean when N > 1)
    torch.cuda.synchronize()  # every stream
    first = float(first_t) * (inv_global_rows if dist_on else 1.0)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    graphs = None
    if args.graph_steps:  # one captured step per rotating input set, replayed in the same order
        graphs = []
        for k in range(len(sets)):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                out = step(k)
            graphs.append(gph)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
    for i in range(args.steps):
        if graphs is not None:
            graphs[i % len(graphs)].replay()
        else:
            out = step(i, events[i] if i % EVENT_STRIDE == 0 else None)
    host_enqueue = time.perf_counter() - t0   # when the host is done issuing work (diagnostic: is the loop host-bound?)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    del out
    leave_lanes()  # back to the default stream for the secondary measurements

    if graphs is not None:  # events cannot sit inside a replay: time the dominant kernel eagerly after the timed region
        for i in range(min(args.steps, 50)):
            step(i, events[i])
        torch.cuda.synchronize()
        events = events[:min(args.steps, 50)]
    if graphs is None:
        events = events[::EVENT_STRIDE]  # the launches that were bracketed inside the timed region
    kern_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)

    def timed(fn, n):  # secondary measurements (outside the contract's timed region), HIP events on the launch stream
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    extras = {"host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps}
    n_extra = max(10, min(args.steps, 50))
    if dist_on:  # BASELINE config 3: report the step with and without the collective
        def local_only(i):
            with torch.no_grad():
                x2, y2, xp, yp, flags, plan, _ = marshalled[i % len(sets)]
                nat.reduce_mean(nat.forward_rows(x2, y2, xp, yp, float(mod.p), flags, plan), sum_out=ring[i % len(ring)])
        extras["ms_per_step_without_collective"] = timed(local_only, n_extra)
    else:
        cut = Wasserstein1D(**MODES["cutoff"]).to(dev)
        ys = [s_[1].clone().requires_grad_(True) for s_ in sets[:2]]

        def fwd_cutoff(i):
            with torch.no_grad():
                cut(sets[i % len(sets)][0], sets[i % len(sets)][1], x_pos=pos_x, y_pos=pos_y)

        def fwd_bwd_cutoff(i):
            yv = ys[i % 2]
            yv.grad = None
            cut(sets[i % 2][0], yv, x_pos=pos_x, y_pos=pos_y).backward()

        cm = [cut._marshal(x, y, pos_x, pos_y, {}) for x, y in sets[:2]]
        one = torch.ones(1, device=dev)

        def fwd_bwd_cutoff_kernels(i):  # the kernels of the autograd step above through the FFI alone (no autograd bookkeeping)
            x2, y2, xp, yp, flags, plan, _ = cm[i % 2]
            with torch.no_grad():
                nat.reduce_mean(nat.forward_rows(x2, y2, xp, yp, float(cut.p), flags, plan))
                nat.backward_rows(x2, y2, xp, yp, float(cut.p), flags, one, need_gx=False, plan=plan, grad_scale=1.0 / B)

        def loss_and_grad_cutoff(i):  # the training form: loss and d loss / d y out of one pass over the rows (+ the mean kernel)
            x2, y2, xp, yp, flags, plan, _ = cm[i % 2]
            with torch.no_grad():
                nat.loss_and_grad(x2, y2, xp, yp, float(cut.p), flags, plan)

        extras["paper_cutoff_mode_forward_ms_per_step"] = timed(fwd_cutoff, n_extra)
        # the module through autograd: the gradient w.r.t. y comes out of the same pass as the loss
        extras["paper_cutoff_mode_forward_backward_ms_per_step"] = timed(fwd_bwd_cutoff, n_extra)
        extras["paper_cutoff_mode_forward_backward_kernels_ms_per_step"] = timed(fwd_bwd_cutoff_kernels, n_extra)  # separate forward, mean, backward
        extras["paper_cutoff_mode_loss_and_grad_ms_per_step"] = timed(loss_and_grad_cutoff, n_extra)
        del ys
    bytes_per_row = 4 * (N + N) + 4  # fwd, shared positions
    achieved = bytes_per_row * B / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        parity = None
        try:
            man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
            if (B, N) == (8192, 2048) and world == 1:  # the stored scalar is that of seed 1234 alone
                want = man["_config2_b8192n2048_seed1234"][f"uniform_{args.mode}"]
                parity = abs(first - want) / abs(want)
        except Exception:
            parity = None
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), if present
            pj = json.load(open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")))
            if pj.get("workload") == f"B={B},N={N},{args.mode}":
                traffic = pj["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        rec = {
            "metric": "sot_loss_evals_per_sec", "value": world * B * args.steps / elapsed, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"SOT-2048 config: B={B} rows/GPU x N_fft={N} fp32 spectrum pairs, forward, mode {args.mode} "
                                   f"({json.dumps(MODES[args.mode])}), shared linspace positions, {len(sets)} rotating input sets "
                                   f"({len(sets) * 2 * B * N * 4 / 2**20:.0f} MiB > 256 MiB L3)",
                       "rows_per_gpu": B, "n_fft": N, "mode": args.mode, "prewarm_steps": args.prewarm, "global_rows": world * B,
                       "streams": n_lanes,
                       "collective": "none" if not dist_on else "one RCCL all-reduce(SUM) of the fp64 partial sum per step" + (" (HIP-graph replay)" if args.graph_steps else ""),
                       "parity_rel_err_vs_reference_scalar": parity, "loss_set0": first},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "sot_forward_full_kernel",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_row * B},
            "extras": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args.mode, N, min(args.cpu_rows, B), 1234)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    barrier()
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
