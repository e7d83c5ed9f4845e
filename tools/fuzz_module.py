"""Randomised sweep of the drop-in MODULE on the GPU box: Wasserstein1D on GPU tensors (HIP kernels through every host route: C++ host
path, hot-call cache, Python binding, early gradient, row losses, `dims`, hinge, 3-D inputs, fixed_x / explicit / per-row positions)
against the same module on CPU tensors (the package's torch-op route, which reproduces the reference bit for bit).
    python tools/fuzz_module.py [seconds=60] [seed=0]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sot_amd.losses import Wasserstein1D

dev = torch.device("cuda:0")
LENGTHS = [5, 33, 64, 129, 200, 257, 300, 512, 513, 700, 1024, 1025, 1100, 2048]


def run(budget=60.0, seed0=0, max_cases=None, verbose=True, grad_tol=2e-3):
    rng = np.random.default_rng(seed0)
    failures, cases, worst_l, worst_g = [], 0, 0.0, 0.0
    t_end = time.time() + budget
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        seed = int(rng.integers(0, 2 ** 31 - 1))
        g = torch.Generator().manual_seed(seed)
        N = int(rng.choice(LENGTHS))
        three_d = rng.random() < 0.4
        lead = (int(rng.integers(1, 5)), int(rng.integers(1, 7))) if three_d else (int(rng.integers(1, 40)),)
        cutoff = rng.random() < 0.3
        p = int(rng.choice([1, 2, 3]))
        square, dn = bool(rng.random() < 0.5), bool(cutoff or rng.random() < 0.2)
        hinge_on = rng.random() < 0.2
        pos_kind = str(rng.choice(["fixed", "shared", "shared_unsorted", "rows"], p=[0.4, 0.3, 0.15, 0.15]))
        ctor = dict(p=p, square_dist=square, dont_normalize=dn, limit_quantile_range=cutoff, hinge=hinge_on)
        if pos_kind == "fixed":
            ctor["fixed_x"] = N
        kwargs = {}
        if hinge_on:
            kwargs["hinge"] = float(rng.choice([0.0, 1e-3, 0.05]))
        if rng.random() < 0.3:
            kwargs["dims"] = int(rng.integers(0, len(lead)))
        if cutoff:     # dyadic weights: the row mass and the cutoff do not depend on the summation order
            mk = lambda: (torch.randint(0, 32, lead + (N,), generator=g).float() / 32)     # noqa: E731
        else:
            # bounded away from zero: a weight below float32's resolution of the CDF makes two levels tie exactly, and the gradient at a
            # tie is a convention on which float32 torch, float64 torch and the kernels legitimately differ
            mk = lambda: 0.05 + torch.rand(lead + (N,), generator=g) ** 3                     # noqa: E731
        if pos_kind == "fixed":
            pos_args = {}
        elif pos_kind == "shared":
            pz = torch.sort(torch.rand(N, generator=g))[0]
            pos_args = dict(x_pos=pz, y_pos=pz.clone())
        elif pos_kind == "shared_unsorted":
            pos_args = dict(x_pos=torch.rand(N, generator=g), y_pos=torch.rand(N, generator=g))
        else:
            pos_args = dict(x_pos=torch.rand(lead + (N,), generator=g), y_pos=torch.rand(lead + (N,), generator=g))
        gx, gy = bool(rng.random() < 0.3), bool(rng.random() < 0.7)
        gp = bool(pos_kind != "fixed" and rng.random() < 0.35)   # round 4: gradients w.r.t. the positions (sot_w1d_position_grad)
        if gp:   # DISTINCT positions: between equal positions torch.sort's (unstable) order decides which of them receives the gradient
            def distinct(shape):
                base = (torch.arange(N)[None, :] + 0.9 * torch.rand((int(np.prod(shape[:-1])) if len(shape) > 1 else 1, N), generator=g)) / N
                if pos_kind != "shared":
                    base = torch.stack([r[torch.randperm(N, generator=g)] for r in base])
                return base.reshape(shape)
            shape = (N,) if pos_kind != "rows" else lead + (N,)
            pos_args = dict(x_pos=distinct(shape), y_pos=distinct(shape))
        want_grad = (gx or gy) and not cutoff          # (dyadic rows tie exactly: the gradient's tie order is a convention)
        desc = dict(seed=seed, N=N, lead=lead, ctor=ctor, kwargs=kwargs, pos=pos_kind, gx=gx, gy=gy, gp=gp)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cpu_mod, gpu_mod = Wasserstein1D(**ctor), Wasserstein1D(**ctor).to(dev)
            ok = True
            for rep in range(3):        # the same module three times: cold call, hot call, hot call on new tensors
                x, y = mk(), mk()
                xc, yc = x.clone().requires_grad_(gx and want_grad), y.clone().requires_grad_(gy and want_grad)
                xg, yg = x.to(dev).requires_grad_(gx and want_grad), y.to(dev).requires_grad_(gy and want_grad)
                pos_gpu = {k: v.to(dev) for k, v in pos_args.items()} if rep == 0 or rng.random() < 0.5 else pos_gpu
                pos_cpu = pos_args
                if gp:   # fresh leaves on both sides (a level tie moves no position gradient: the second member of a tie has zero width)
                    pos_cpu = {k: v.clone().requires_grad_(True) for k, v in pos_args.items()}
                    pos_gpu = {k: v.to(dev).requires_grad_(True) for k, v in pos_args.items()}
                want = cpu_mod(xc, yc, **pos_cpu, **kwargs)
                got = gpu_mod(xg, yg, **pos_gpu, **kwargs)
                if got.shape != want.shape:
                    failures.append(("SHAPE", desc)); verbose and print("SHAPE", desc, tuple(got.shape), tuple(want.shape)); ok = False; break
                scale = float(want.detach().abs().max()) + float(kwargs.get("hinge", 0.0)) + 1e-12   # (relu(row - hinge) cancels: errors scale with the row loss)
                el = float((got.detach().cpu() - want.detach()).abs().max()) / scale
                worst_l = max(worst_l, el)
                if not el <= 3e-5:
                    failures.append(("LOSS", desc, el)); verbose and print("LOSS", desc, "rep", rep, "err", el); ok = False; break
                if (want_grad or gp) and want.requires_grad:
                    w = torch.rand(want.shape, generator=g)
                    (want * w).sum().backward()
                    (got * w.to(dev)).sum().backward()
                    if gp:
                        for name in ("x_pos", "y_pos"):
                            a, b = pos_gpu[name].grad, pos_cpu[name].grad
                            gs = float(b.abs().max()) + 1e-30
                            eg = float((a.cpu() - b).abs().max()) / gs
                            worst_g = max(worst_g, eg) if eg <= grad_tol else worst_g
                            if not eg <= 2e-4:
                                failures.append(("PGRAD", desc, name, eg)); verbose and print("PGRAD", name, desc, "rep", rep, "err / max", eg); ok = False
                    for name, a, b in (("x", xg, xc), ("y", yg, yc)):
                        if b.grad is None:
                            continue
                        gs = float(b.grad.abs().max()) + 1e-30
                        eg = float((a.grad.cpu() - b.grad).abs().max()) / gs
                        worst_g = max(worst_g, eg)
                        if not eg <= grad_tol:
                            failures.append(("GRAD", desc, name, eg)); verbose and print("GRAD", name, desc, "rep", rep, "err / max", eg); ok = False
                    if not ok:
                        break
        cases += 1
    if verbose:
        print(f"cases {cases}, failures {len(failures)}, worst loss err {worst_l:.3g}, worst gradient err / max {worst_g:.3g}")
    return cases, failures, worst_l, worst_g


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
