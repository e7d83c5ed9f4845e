"""Randomised parity sweep on the GPU box: the HIP path (every dispatch the C ABI chooses by itself) against the C oracle on random
shapes, position kinds, weight kinds, modes and strides.  Prints every case outside the tests' tolerances with the parameters that
reproduce it.
    python tools/fuzz_gpu.py [seconds=120] [seed=0]
tests/test_gpu_fuzz.py runs a fixed number of cases of the same generator."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from sot_amd import _native as nat
from oracle import sot_oracle as so
from oracle.inputs import gen_inputs

nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
INTERESTING = [1, 2, 3, 7, 8, 9, 63, 64, 65, 127, 128, 129, 130, 255, 256, 257, 258, 511, 512, 513, 514, 1023, 1024, 1025, 1026, 2047, 2048,
               2049, 2050, 4095, 4096, 4097]


def pick_len():
    r = rng.random()
    if r < 0.35:
        return int(rng.choice(INTERESTING))
    if r < 0.85:
        return int(rng.integers(1, 4200))
    return int(rng.integers(4200, 8193))


def positions(kind, n, B, g):
    if kind == "linspace":
        return torch.linspace(0, 1, n)
    if kind == "rfft":
        f = torch.fft.rfftfreq(2 * max(n - 1, 1), 1 / 16000.0)[:n]
        return (f / max(float(f.max()), 1e-9)).float()
    if kind == "sorted":
        return torch.sort(torch.rand(n, generator=g))[0]
    if kind == "unsorted":
        return torch.rand(n, generator=g)
    if kind == "ties":      # repeated positions, unsorted
        return (torch.randint(0, max(2, n // 3), (n,), generator=g).float() / max(2, n // 3))
    if kind == "rows":      # per-row positions
        return torch.rand(B, n, generator=g)
    raise ValueError(kind)


def weights(kind, B, n, m, seed):
    if kind == "sparse":
        x, y = gen_inputs("uniform", B, n, m, seed)
        g = torch.Generator().manual_seed(seed + 1)
        x = x * (torch.rand(B, n, generator=g) < 0.3)
        y = y * (torch.rand(B, m, generator=g) < 0.3)
        return x, y
    return gen_inputs(kind, B, n, m, seed)


def run(budget=120.0, seed0=0, max_cases=None, grad_tol=2e-5, verbose=True):
    """(cases, failures): failures = list of (kind, parameters, error)."""
    global rng
    rng = np.random.default_rng(seed0)
    failures = []
    t_end = time.time() + budget
    cases = bad = 0
    worst_f = worst_b = 0.0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        seed = int(rng.integers(0, 2 ** 31 - 1))
        g = torch.Generator().manual_seed(seed)
        n = pick_len()
        m = n if rng.random() < 0.7 else pick_len()
        pk = str(rng.choice(["linspace", "rfft", "sorted", "unsorted", "ties", "rows"], p=[0.35, 0.15, 0.1, 0.15, 0.1, 0.15]))
        if pk == "rows":
            n, m = min(n, 2100), min(m, 2100)
        B = int(rng.integers(1, max(2, min(48, 200000 // (n + m) + 2))))
        wk = str(rng.choice(["uniform", "peaky", "dyadic", "edge", "sparse"]))
        p = float(rng.choice([1.0, 2.0, 1.5, 3.0], p=[0.35, 0.35, 0.15, 0.15]))
        flags = int(rng.integers(0, 8))            # SQUARE | DONT_NORMALIZE | LIMIT_Q
        same = (m == n) and rng.random() < 0.8
        xpos = positions(pk, n, B, g)
        ypos = xpos.clone() if same else positions(pk, m, B, g)
        sorted_pos = pk in ("linspace", "rfft", "sorted")
        if not sorted_pos or rng.random() < 0.7:
            flags |= nat.FLAG_REQUIRE_SORT
        x, y = weights(wk, B, n, m, seed)
        strided = rng.random() < 0.2
        if strided:     # rows that are views into wider buffers
            bx, by = torch.zeros(B, n + 3), torch.zeros(B, m + 5)
            bx[:, 1:n + 1], by[:, 2:m + 2] = x, y
            xd, yd = bx.to(dev)[:, 1:n + 1], by.to(dev)[:, 2:m + 2]
        else:
            xd, yd = x.to(dev), y.to(dev)
        xpd, ypd = xpos.to(dev), ypos.to(dev)
        use_plan = xpos.ndim == 1 and (flags & nat.FLAG_REQUIRE_SORT) and rng.random() < 0.7
        plan = nat.PositionPlan(xpd, ypd) if use_plan else None
        desc = dict(seed=seed, B=B, n=n, m=m, pos=pk, same=bool(same), w=wk, p=p, flags=flags, plan=bool(use_plan), strided=bool(strided))
        try:
            got = nat.forward_rows(xd, yd, xpd, ypd, p, flags, plan).cpu().numpy()
        except nat.SotError as e:
            if e.status == nat.SOT_ERR_UNSUPPORTED_SIZE:
                continue
            failures.append(("ERROR forward", desc)); verbose and print("ERROR forward", desc, e); bad += 1; continue
        want = so.forward(x.numpy(), y.numpy(), xpos.numpy(), ypos.numpy(), p=p, flags=flags & 15)
        err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-6)))
        worst_f = max(worst_f, err)
        cases += 1
        if not np.isfinite(got).all() or err > 2e-5:
            bad += 1
            failures.append(("FORWARD", desc)); verbose and print("FORWARD", desc, "max rel err", err, "row", int(np.argmax(np.abs(got - want) / np.maximum(np.abs(want), 1e-6))))
        if n + m <= 12000:
            grow = torch.linspace(0.5, 1.5, B)
            try:
                gx, gy = nat.backward_rows(xd, yd, xpd, ypd, p, flags, grow.to(dev), plan=plan, grad_scale=0.5)
            except nat.SotError as e:
                if e.status == nat.SOT_ERR_UNSUPPORTED_SIZE:
                    continue
                failures.append(("ERROR backward", desc)); verbose and print("ERROR backward", desc, e); bad += 1; continue
            wx, wy = so.backward(x.numpy(), y.numpy(), xpos.numpy(), ypos.numpy(), (0.5 * grow).numpy(), p=p, flags=flags & 15)
            # one scale per row for both gradients: entries that are zero up to rounding (cancellation in the normalisation's mass term)
            # are compared against the row's gradient scale, floored by loss / mass (the natural size of d loss / d weight)
            sq = 2 if flags & nat.FLAG_SQUARE else 1
            mass = np.minimum((x.numpy().astype(np.float64) ** sq).sum(1), (y.numpy().astype(np.float64) ** sq).sum(1))
            floor = 1e-5 * (0.5 * grow.numpy()) * np.abs(want) / np.maximum(mass, 1e-7)
            scale = np.maximum(np.maximum(np.abs(wx).max(axis=1), np.abs(wy).max(axis=1)), floor)[:, None] + 1e-30
            for name, gg, ww in (("gx", gx, wx), ("gy", gy, wy)):
                gg = gg.cpu().numpy()
                e = float(np.max(np.abs(gg - ww) / scale))
                worst_b = max(worst_b, e)
                if not np.isfinite(gg).all() or e > grad_tol:
                    bad += 1
                    r = int(np.argmax((np.abs(gg - ww) / scale).max(axis=1)))
                    failures.append(("BACKWARD", desc)); verbose and print("BACKWARD", name, desc, "max err / row gradient scale", e, "row", r, "scale", float(scale[r, 0]))
            if rng.random() < 0.5:   # the training form agrees with forward + y-only backward bit for bit
                mean, rows2, gy2 = nat.loss_and_grad(xd, yd, xpd, ypd, p, flags | nat.FLAG_NO_AREA, plan)
                ones = nat.backward_rows(xd, yd, xpd, ypd, p, flags, torch.ones(1, device=dev), need_gx=False, plan=plan, grad_scale=1.0 / B)[1]
                rows1 = nat.forward_rows(xd, yd, xpd, ypd, p, flags | nat.FLAG_NO_AREA, plan)
                if not (torch.equal(rows1, rows2) and torch.equal(gy2, ones)):
                    bad += 1
                    failures.append(("TRAINING FORM differs from forward + backward", desc)); verbose and print("TRAINING FORM differs from forward + backward", desc)
    if verbose:
        print(f"cases {cases}, outside tolerance {bad}, worst forward rel err {worst_f:.3g}, worst gradient err / row gradient scale {worst_b:.3g}")
    return cases, failures, worst_f, worst_b


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
