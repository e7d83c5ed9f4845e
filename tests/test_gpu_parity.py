"""-m gpu: the HIP path (through the C ABI) against the golden vectors captured from the reference
and against the CPU oracle on identical inputs.

Tolerances (BASELINE north_star: 1e-5 relative fp32, sort indices bit-exact):
  * per-row loss and batch mean: rel <= 1e-5 -- in practice the row mass S and the CDFs are
    bit-identical to the reference's and rows differ by a few ulp of summation order;
  * CDFs U, V / levels Q / quantiles: bit-exact except for fp64-association double roundings
    (allowed: <= 1 ulp on <= 1e-4 of the elements);
  * gradients: <= 1e-5 of the row's max |grad| against the closed-form oracle (stable tie order).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, case_names, ctor_to_flags, load_case
from gpu_util import device, module_for, native, to_dev, ulp_diff

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def pos_kwargs(meta, g):
    if meta["pos"] == "fixed_x":
        return {}
    return dict(x_pos=to_dev(g["x_pos"]), y_pos=to_dev(g["y_pos"]))


def run_rows(mod, x, y, pk):
    """per-row losses via the module (dims keeps every row)"""
    x2 = x.reshape(-1, 1, x.shape[-1])
    y2 = y.reshape(-1, 1, y.shape[-1])
    pk2 = {k: (v.reshape(-1, 1, v.shape[-1]) if v.ndim >= 2 else v) for k, v in pk.items()}
    return mod(x2, y2, dims=[1], **pk2)


@pytest.mark.parametrize("name", case_names())
def test_forward_matches_reference_fixture(name, manifest):
    meta, g = load_case(name, manifest)
    mod = module_for(meta["ctor"])
    x, y = to_dev(g["x"]), to_dev(g["y"])
    pk = pos_kwargs(meta, g)
    scalar = mod(x, y, **pk)
    assert scalar.ndim == 0 and scalar.dtype == torch.float32
    rows = run_rows(mod, x, y, pk).cpu().numpy()
    want_rows = g["row_loss"].reshape(-1)
    np.testing.assert_allclose(rows, want_rows, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(float(scalar), float(g["scalar"]), rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize("name", [n for n in case_names() if n.startswith(("b4n512", "nm_", "unsorted", "edge", "fixedx"))])
def test_quantile_tensors_match_reference(name, manifest):
    meta, g = load_case(name, manifest)
    mod = module_for(meta["ctor"])
    x, y = to_dev(g["x"]), to_dev(g["y"])
    out = mod(x, y, return_quantiles=True, **pos_kwargs(meta, g))
    lead = tuple(g["x"].shape[:-1])
    for got, key in zip(out, ("uq", "vq", "Q", "U", "V")):
        assert tuple(got.shape[:-1]) == lead
        got = got.cpu().numpy().reshape(g[key].shape)
        d = ulp_diff(got, g[key])
        if key in ("U", "V", "Q"):
            assert d.max() <= 1 and (d > 0).mean() <= 1e-4, (key, d.max(), (d > 0).mean())
        else:  # quantile values: a 1-ulp CDF flip can move a rank by one position
            assert (got != g[key]).mean() <= 1e-3, key


@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("seeded_b256")])
def test_backward_matches_oracle(name, manifest):
    from oracle import sot_oracle as so
    meta, g = load_case(name, manifest)
    p, flags = ctor_to_flags(meta["ctor"])
    mod = module_for(meta["ctor"])
    x = to_dev(g["x"]).requires_grad_(True)
    y = to_dev(g["y"]).requires_grad_(True)
    loss = mod(x, y, **pos_kwargs(meta, g))
    loss.backward()
    x2 = g["x"].reshape(-1, g["x"].shape[-1])
    y2 = g["y"].reshape(-1, g["y"].shape[-1])
    xp = g["x_pos"].reshape(-1, g["x_pos"].shape[-1]) if g["x_pos"].ndim == 3 else g["x_pos"]
    yp = g["y_pos"].reshape(-1, g["y_pos"].shape[-1]) if g["y_pos"].ndim == 3 else g["y_pos"]
    B = x2.shape[0]
    gx, gy = so.backward(x2, y2, xp, yp, np.full(B, 1.0 / B, np.float32), p=p, flags=flags)
    for got, want in ((x.grad, gx), (y.grad, gy)):
        got = got.cpu().numpy().reshape(want.shape)
        tol = 1e-5 * np.abs(want).max(axis=1, keepdims=True) + 3e-8
        assert (np.abs(got - want) <= tol).all(), (name, np.abs(got - want).max())


@pytest.mark.parametrize("kind", ["uniform", "peaky", "dyadic"])
@pytest.mark.parametrize("mode", ["p1", "cutoff", "nocut"])
def test_config2_full_size_scalars(kind, mode, manifest):
    """BASELINE config 2: B=8192, N=2048, inputs regenerated from the seed; scalar from the reference."""
    from oracle.inputs import gen_inputs, sha256_of
    from oracle.make_golden import MODES
    big = manifest["_config2_b8192n2048_seed1234"]
    x, y = gen_inputs(kind, 8192, 2048, 2048, 1234)
    assert sha256_of(x, y) == big[f"{kind}_sha256"]
    pos = torch.linspace(0, 1, 2048).to(device())
    mod = module_for(MODES[mode])
    got = float(mod(x.to(device()), y.to(device()), x_pos=pos, y_pos=pos.clone()))
    want = big[f"{kind}_{mode}"]
    assert abs(got - want) <= RTOL * abs(want), (got, want)


@pytest.mark.parametrize("mode", ["p1", "cutoff"])
def test_config3_global_batch_on_one_gpu(mode):
    """BASELINE config 3 (B = 65536 = 8 ranks x 8192 rows x 2048 bins, rank r seeded 1234 + r) evaluated on ONE GPU the way the 8 ranks
    evaluate it: per shard the module's row losses and their fixed-order fp64 sum (distributed._RowSum = sot_w1d_reduce_mean's sum_out),
    the 8 (sum, rows) pairs added as the all-reduce(SUM) adds them (distributed._AllReduceSumCount), mean = sum / rows as float32 --
    against the OpenMP oracle's rows and mean over all 65 536 rows.  (No 8-GPU node has been available to the driver: the arithmetic of the
    reduction is what can be pinned without one; the collective itself runs under RCCL at world size 1, tests/test_rccl_gpu.py.)"""
    from oracle import sot_oracle as so
    from oracle.make_golden import MODES
    from sot_amd import distributed as sd
    from sot_amd.bench_inputs import spectrum_pairs
    dev = device()
    mod = module_for(MODES[mode])
    pos = torch.linspace(0, 1, 2048)
    pos_d, pos_d2 = pos.to(dev), pos.to(dev).clone()
    flags = so.make_flags(MODES[mode].get("square_dist", False), MODES[mode].get("dont_normalize", False),
                          MODES[mode].get("limit_quantile_range", False), True)
    packed = torch.zeros(2, dtype=torch.float64, device=dev)
    want_rows = []
    for r in range(8):
        x, y = spectrum_pairs("uniform", 8192, 2048, 2048, 1234 + r)
        rows = mod.row_losses(x.to(dev), y.to(dev), x_pos=pos_d, y_pos=pos_d2)
        local_sum = sd._RowSum.apply(rows.float().contiguous())
        packed += torch.stack([local_sum.to(torch.float64).reshape(()), torch.tensor(float(rows.numel()), dtype=torch.float64, device=dev)])
        want = so.forward(x.numpy(), y.numpy(), pos.numpy(), pos.numpy(), p=float(MODES[mode].get("p", 1)), flags=flags)
        np.testing.assert_allclose(rows.cpu().numpy(), want, rtol=RTOL)          # every row of every shard
        want_rows.append(want)
    got = float((packed[0] / packed[1]).to(torch.float32))
    want_mean = float(so.mean(np.concatenate(want_rows)))
    assert int(packed[1]) == 65536
    assert abs(got - want_mean) <= 1e-6 * abs(want_mean), (got, want_mean)
    # shard 0 is config 2's batch: its own mean is the reference's stored scalar
    # (tests/golden/manifest.json: _config2_b8192n2048_seed1234), so the global figure hangs on a reference value, not only on the oracle
    assert abs(float(so.mean(want_rows[0])) - json.load(open(os.path.join(GOLDEN, "manifest.json")))["_config2_b8192n2048_seed1234"][f"uniform_{mode}"]) \
        <= 1e-6 * abs(float(so.mean(want_rows[0])))


@pytest.mark.parametrize("N,B", [(512, 1030), (1024, 777), (2048, 520), (4096, 130), (8192, 70)])
@pytest.mark.parametrize("flags,p", [(0, 1.0), (1, 1.0), (1 | 2 | 4, 2.0), (4, 1.0), (1 | 4 | 8, 2.0), (2, 2.0), (0, 3.0), (1 | 2 | 4, 1.5)])
def test_full_row_kernel_matches_generic(N, B, flags, p):
    """Rows that fill their launch geometry (n == m == G*CPT) run the fully specialised forward kernel: it must agree
    bit for bit with the generic kernel (SOT_FLAG_NO_SPECIALIZE) and, on EVERY row, with the oracle.  (4096 bins:
    the generic kernel runs 256 threads x 16 elements, the specialised one 512 x 8: the same terms, another association of
    the row's final sum, hence a few ulp.)"""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    x, y = gen_inputs("peaky", B, N, N, 4321 + N)
    x, y = x.to(device()), y.to(device())
    pos = torch.linspace(0, 1, N).to(device())
    pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2) if flags & 8 else None
    spec = nat.forward_rows(x, y, pos, pos2, p, flags, plan)
    gen = nat.forward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, plan)
    if N == 4096 or p not in (1.0, 2.0):   # general p (round 3: compile-time-length kernels with powf in the walk): same terms, the
        torch.testing.assert_close(spec, gen, rtol=2e-6, atol=1e-12)   # loser-form walk may evaluate |a - b| as |b - a| -> same value
    else:
        assert torch.equal(spec, gen), float((spec - gen).abs().max())
    k = B   # every row against the oracle (round 4: the oracle runs its rows on all host cores, oracle/Makefile -fopenmp)
    want = so.forward(x[:k].cpu().numpy(), y[:k].cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(), p=p, flags=flags & 15)
    np.testing.assert_allclose(spec[:k].cpu().numpy(), want, rtol=RTOL)


@pytest.mark.parametrize("N,B", [(512, 530), (1024, 301), (2048, 200), (4096, 67)])
@pytest.mark.parametrize("flags,p", [(0, 1.0), (1, 1.0), (1 | 2 | 4, 2.0), (4, 1.0), (1 | 4 | 8, 2.0), (2, 2.0), (0, 3.0), (1 | 2 | 4, 1.5)])
def test_full_row_backward_matches_generic_and_oracle(N, B, flags, p):
    """The specialised backward kernel (rows with n == m == 512 / 2048) gives bit for bit the gradients of the generic
    kernel, and the oracle's closed form on every row."""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    x, y = gen_inputs("peaky", B, N, N, 977 + N)
    x, y = x.to(device()), y.to(device())
    pos = torch.linspace(0, 1, N).to(device())
    pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2) if flags & 8 else None
    g = torch.linspace(0.5, 1.5, B).to(device())
    sx, sy = nat.backward_rows(x, y, pos, pos2, p, flags, g, plan=plan, grad_scale=0.25)
    gx, gy = nat.backward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, g, plan=plan, grad_scale=0.25)
    if p in (1.0, 2.0):
        assert torch.equal(sx, gx) and torch.equal(sy, gy), (float((sx - gx).abs().max()), float((sy - gy).abs().max()))
    else:   # general p: the same closed form on powf costs
        for got, ref in ((sx, gx), (sy, gy)):
            assert float(((got - ref).abs() / (ref.abs().amax(dim=1, keepdim=True) + 1e-30)).max()) <= 2e-6
    only_y = nat.backward_rows(x, y, pos, pos2, p, flags, g, need_gx=False, plan=plan, grad_scale=0.25)
    assert only_y[0] is None and torch.equal(only_y[1], sy)
    k = B   # every row against the oracle
    wx, wy = so.backward(x[:k].cpu().numpy(), y[:k].cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(),
                         (0.25 * g[:k]).cpu().numpy(), p=p, flags=flags & 15)
    for got, want in ((sx[:k].cpu().numpy(), wx), (sy[:k].cpu().numpy(), wy)):
        scale = np.abs(want).max(axis=1, keepdims=True) + 1e-30
        assert np.max(np.abs(got - want) / scale) <= 1e-5


@pytest.mark.parametrize("N,B", [(512, 530), (1024, 301), (2048, 200), (4096, 67), (129, 700), (257, 531), (513, 300), (1025, 261), (2049, 133)])
@pytest.mark.parametrize("flags", [8, 8 | 1, 8 | 2, 8 | 1 | 2])
@pytest.mark.parametrize("kind", ["peaky", "uniform", "permuted"])
def test_merge_free_training_form_p1(N, B, flags, kind):
    """OPT-IN (SOT_FLAG_TIE_FREE_GRADIENT; Wasserstein1D(..., tie_free_gradient=True)): p = 1 on one grid, gradient w.r.t. y alone, from
    sot_area_train_kernel (round 4) -- no merge, one compare per element and an fp64 suffix scan.  It returns the derivative of
    loss = sum_i |U_i - V_i| (x_{i+1} - x_i) in the float32 CDF values, i.e. what autograd gives for that formula on the reference's own
    CDFs (checked on every row at 1e-5 of the row's largest entry) and what float64 autograd of the reference gives; the reference's
    float32 autograd (and the oracle, and the merge backward kernel, which reproduce its stable tie order) differs from it only on rows
    with exactly tied levels -- checked: on rows WITHOUT any tie all three agree.  The row losses are the merge-free forward kernel's bit
    for bit; without the flag the call runs the merge backward as before."""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    x, y = gen_inputs("peaky" if kind != "uniform" else "uniform", B, N, N, 1977 + N)
    x, y = x.to(device()), y.to(device())
    pos = torch.linspace(0, 1, N)
    if kind == "permuted":
        pos = pos[torch.randperm(N, generator=torch.Generator().manual_seed(N))]
    pos = pos.to(device()); pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2)
    assert plan.same_grid()
    TF = nat.FLAG_TIE_FREE_GRADIENT
    g = torch.linspace(0.5, 1.5, B).to(device())
    _, gy = nat.backward_rows(x, y, pos, pos2, 1.0, flags | TF, g, need_gx=False, plan=plan, grad_scale=0.25)   # merge-free
    _, gm = nat.backward_rows(x, y, pos, pos2, 1.0, flags, g, need_gx=False, plan=plan, grad_scale=0.25)        # merge walk (default)
    _, wy = so.backward(x.cpu().numpy(), y.cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(), (0.25 * g).cpu().numpy(), p=1.0, flags=flags & 15)
    scale = np.abs(wy).max(axis=1, keepdims=True) + 1e-30
    assert np.max(np.abs(gm.cpu().numpy() - wy) / scale) <= 1e-5, "the default (merge backward) keeps the reference's tie order"
    # autograd of the area formula on the reference's own float32 CDFs (torch CPU: the same bits as the kernel's)
    xc, yc = x.cpu(), y.cpu().requires_grad_(True)
    ps, order = torch.sort(pos.cpu())
    sq, dn = bool(flags & 1), bool(flags & 2)
    xa, ya = (xc ** 2, yc ** 2) if sq else (xc, yc)
    mass_x = xa.sum(1, keepdim=True)
    eps = torch.tensor(1e-7)
    a = xa / torch.where(mass_x <= 1e-7, eps, mass_x)
    mass_y = mass_x if dn else ya.sum(1, keepdim=True)
    bb = ya / torch.where(mass_y <= 1e-7, eps, mass_y)
    U, V = torch.cumsum(a[:, order], 1), torch.cumsum(bb[:, order], 1)
    loss = ((U - V).abs()[:, :-1] * (ps[1:] - ps[:-1])).sum(1)
    (loss * 0.25 * g.cpu()).sum().backward()
    want = yc.grad.numpy()
    wscale = np.abs(want).max(axis=1, keepdims=True) + 1e-30
    assert np.max(np.abs(gy.cpu().numpy() - want) / wscale) <= 1e-5, "merge-free training form vs autograd of the area formula"
    # rows without any tied level (within U, within V, across): every convention gives the same gradient
    lv = torch.sort(torch.cat((U, V), 1).detach(), dim=1)[0]
    clean = (lv[:, 1:-1] != lv[:, :-2]).all(dim=1).numpy()   # (the last two levels, U_last and V_last, often are both 1.0: cell width 0 there)
    if clean.any():   # (exact coincidences U_i == V_j grow like N^2 / 2^24 per row, squares push small weights below the CDF's resolution)
        assert np.max((np.abs(gy.cpu().numpy() - wy) / scale)[clean]) <= 2e-5, "tie-free rows: merge-free form == oracle"
    mean, rows, gy2 = nat.loss_and_grad(x, y, pos, pos2, 1.0, flags | TF, plan)
    fwd = nat.forward_rows(x, y, pos, pos2, 1.0, flags, plan)
    if N in (129, 1025):   # the forward runs two rows per wave / one wave per row there: another grouping of the thread-local sums
        torch.testing.assert_close(rows, fwd, rtol=2e-6, atol=1e-12)
    else:
        assert torch.equal(rows, fwd)
    _, gy3 = nat.backward_rows(x, y, pos, pos2, 1.0, flags | TF, torch.ones(1, device=device()), need_gx=False, plan=plan, grad_scale=1.0 / B)
    assert torch.equal(gy2, gy3)
    assert abs(float(mean) - float(rows.double().mean())) <= 1e-6 * abs(float(mean)) + 1e-12


@pytest.mark.gpu
def test_module_tie_free_gradient_option():
    """Wasserstein1D(p=1, tie_free_gradient=True): same loss as the default module bit for bit, gradients equal on rows without ties, the
    training step runs sot_area_train_kernel (timing is in bench.py); the default module is untouched."""
    from sot_amd.losses import Wasserstein1D
    native()
    g = torch.Generator(device=device()).manual_seed(9)
    x = torch.rand(64, 2048, device=device(), generator=g) + 0.05
    y0 = torch.rand(64, 2048, device=device(), generator=g) + 0.05
    pos = torch.linspace(0, 1, 2048, device=device()); pos2 = pos.clone()
    outs = []
    for opt in (False, True):
        mod = Wasserstein1D(p=1, tie_free_gradient=opt).to(device())
        y = y0.clone().requires_grad_(True)
        loss = mod(x, y, x_pos=pos, y_pos=pos2)
        loss.backward()
        outs.append((loss.detach(), y.grad))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=2e-6, atol=0)   # the default training step accumulates the loss on the merge walk
    d = (outs[0][1] - outs[1][1]).abs().amax(dim=1) / outs[0][1].abs().amax(dim=1)
    assert float(d.median()) <= 1e-5   # most rows of bounded-away-from-zero weights have no exact tie
    assert float(d.max()) <= 5e-2


@pytest.mark.parametrize("N,B", [(129, 1500), (257, 1031), (513, 300), (1025, 261), (2049, 133)])
@pytest.mark.parametrize("flags,p", [(0, 1.0), (1, 1.0), (1 | 2 | 4, 2.0), (4, 1.0), (1 | 2 | 4 | 8, 2.0), (2, 2.0), (0, 3.0), (1 | 2 | 4, 1.5)])
@pytest.mark.parametrize("kind", ["peaky", "uniform"])
def test_paper_row_lengths_compile_time_kernel(N, B, flags, p, kind):
    """n_fft 512 / 1024 / 2048 -> 257 / 513 / 1025 bins run a forward kernel with the row length at compile time (other
    thread geometry than the generic kernel: same arithmetic per element, different association of the final sum): it
    must agree with the generic kernel to a few ulp and with the oracle; rows start at arbitrary alignments."""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    x, y = gen_inputs(kind, B, N, N, 55 + N)
    x, y = x.to(device()), y.to(device())
    pos = torch.fft.rfftfreq(2 * (N - 1), 1 / 16000.0)
    pos = (pos / pos.max()).float().to(device())
    pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2) if flags & 8 else None
    spec = nat.forward_rows(x, y, pos, pos2, p, flags, plan)
    gen = nat.forward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, plan)
    torch.testing.assert_close(spec, gen, rtol=2e-6, atol=1e-12)
    k = B   # every row against the oracle
    want = so.forward(x[:k].cpu().numpy(), y[:k].cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(), p=p, flags=flags & 15)
    np.testing.assert_allclose(spec[:k].cpu().numpy(), want, rtol=RTOL)


@pytest.mark.parametrize("N,B", [(129, 90), (257, 70), (513, 37), (1025, 29), (2049, 21)])
@pytest.mark.parametrize("flags,p", [(0, 1.0), (1 | 2 | 4, 2.0), (1 | 4 | 8, 2.0), (2, 1.0), (0, 3.0), (1 | 2 | 4, 1.5)])
def test_paper_row_lengths_backward(N, B, flags, p):
    """Backward kernel with the row length at compile time (257 / 513 / 1025 bins): gradients equal to the generic kernel's
    (same closed form; the fp64 suffix sums are associated differently) and to the oracle's."""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    x, y = gen_inputs("peaky", B, N, N, 311 + N)
    x, y = x.to(device()), y.to(device())
    pos = torch.linspace(0, 1, N).to(device()); pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2) if flags & 8 else None
    g = torch.linspace(0.5, 1.5, B).to(device())
    sx, sy = nat.backward_rows(x, y, pos, pos2, p, flags, g, plan=plan, grad_scale=0.5)
    gx, gy = nat.backward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, g, plan=plan, grad_scale=0.5)
    k = B   # every row against the oracle
    wx, wy = so.backward(x[:k].cpu().numpy(), y[:k].cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(),
                         (0.5 * g[:k]).cpu().numpy(), p=p, flags=flags & 15)
    for got, ref, want in ((sx, gx, wx), (sy, gy, wy)):
        scale = ref.abs().amax(dim=1, keepdim=True) + 1e-30
        assert float(((got - ref).abs() / scale).max()) <= 2e-6
        wscale = np.abs(want).max(axis=1, keepdims=True) + 1e-30
        assert np.max(np.abs(got[:k].cpu().numpy() - want) / wscale) <= 1e-5
    only_y = nat.backward_rows(x, y, pos, pos2, p, flags, g, need_gx=False, plan=plan, grad_scale=0.5)
    assert only_y[0] is None and torch.equal(only_y[1], sy)


def test_paper_row_lengths_unsorted_positions_and_strides():
    nat = native()
    from oracle.inputs import gen_inputs
    for N, B in ((257, 9), (1025, 5)):
        x, y = gen_inputs("uniform", B, N, N, 7 + N)
        xs = torch.zeros(B, N + 3, device=device()); ys = torch.zeros(B, N + 3, device=device())
        xs[:, :N] = x.to(device()); ys[:, :N] = y.to(device())
        xv, yv = xs[:, :N], ys[:, :N]
        g = torch.Generator().manual_seed(N)
        pos = torch.linspace(0, 1, N)[torch.randperm(N, generator=g)].to(device())
        pos2 = torch.linspace(0, 3, N)[torch.randperm(N, generator=g)].to(device())
        for flags, p in [(8, 1.0), (1 | 2 | 4 | 8, 2.0)]:
            spec = nat.forward_rows(xv, yv, pos, pos2, p, flags)
            gen = nat.forward_rows(xv, yv, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE)
            torch.testing.assert_close(spec, gen, rtol=2e-6, atol=1e-12)
            one = torch.ones(1, device=device())
            sx, sy = nat.backward_rows(xv, yv, pos, pos2, p, flags, one, grad_scale=1.0 / B)
            gx, gy = nat.backward_rows(xv, yv, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, one, grad_scale=1.0 / B)
            for got, ref in ((sx, gx), (sy, gy)):
                scale = ref.abs().amax(dim=1, keepdim=True) + 1e-30
                assert float(((got - ref).abs() / scale).max()) <= 2e-6


@pytest.mark.parametrize("N,B", [(2048, 1), (2048, 3), (512, 5), (8192, 2)])
def test_full_row_kernels_strided_rows_and_tiny_batches(N, B):
    """Row strides larger than the row length, batches smaller than a workgroup's row count, broadcast upstream gradient."""
    from oracle.inputs import gen_inputs
    nat = native()
    x, y = gen_inputs("uniform", B, N, N, 31 + B)
    pad = 64
    xs = torch.zeros(B, N + pad, device=device())
    ys = torch.zeros(B, N + pad, device=device())
    xs[:, :N] = x.to(device()); ys[:, :N] = y.to(device())
    xv, yv = xs[:, :N], ys[:, :N]          # non-contiguous views: row stride N + 64
    pos = torch.linspace(0, 1, N).to(device()); pos2 = pos.clone()
    for flags, p in [(0, 1.0), (1 | 2 | 4, 2.0)]:
        pr_spec = nat.forward_rows(xv, yv, pos, pos2, p, flags)
        pr_gen = nat.forward_rows(xv, yv, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE)
        pr_contig = nat.forward_rows(xv.contiguous(), yv.contiguous(), pos, pos2, p, flags)
        assert torch.equal(pr_spec, pr_gen) and torch.equal(pr_spec, pr_contig)
        if N <= 2048:
            one = torch.ones(1, device=device())
            sx, sy = nat.backward_rows(xv, yv, pos, pos2, p, flags, one, grad_scale=1.0 / B)
            gx, gy = nat.backward_rows(xv, yv, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, one, grad_scale=1.0 / B)
            assert torch.equal(sx, gx) and torch.equal(sy, gy)


def test_full_row_kernel_with_unsorted_shared_positions():
    """The specialised kernel gathers through the shared sort permutation when the positions arrive unsorted."""
    nat = native()
    from oracle.inputs import gen_inputs
    N, B = 2048, 64
    x, y = gen_inputs("uniform", B, N, N, 99)
    x, y = x.to(device()), y.to(device())
    g = torch.Generator().manual_seed(5)
    pos = torch.linspace(0, 1, N)[torch.randperm(N, generator=g)].to(device())
    pos2 = torch.linspace(0, 2, N)[torch.randperm(N, generator=g)].to(device())
    for flags, p in [(8, 1.0), (1 | 2 | 4 | 8, 2.0)]:
        spec = nat.forward_rows(x, y, pos, pos2, p, flags)
        gen = nat.forward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE)
        assert torch.equal(spec, gen)
        g = torch.ones(B, device=device())
        sx, sy = nat.backward_rows(x, y, pos, pos2, p, flags, g)
        gx, gy = nat.backward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, g)
        assert torch.equal(sx, gx) and torch.equal(sy, gy)


def test_full_size_properties():
    """Size-independent properties at B=8192, N=2048 (no oracle needed)."""
    from oracle.inputs import gen_inputs
    x, y = gen_inputs("peaky", 8192, 2048, 2048, 77)
    x, y = x.to(device()), y.to(device())
    pos = torch.linspace(0, 1, 2048).to(device())
    w1 = module_for(dict(p=1))
    rows_xy = run_rows(w1, x, y, dict(x_pos=pos, y_pos=pos))
    rows_yx = run_rows(w1, y, x, dict(x_pos=pos, y_pos=pos))
    torch.testing.assert_close(rows_xy, rows_yx, rtol=2e-6, atol=0)          # symmetry
    rows_sc = run_rows(w1, 3.0 * x, 0.25 * y, dict(x_pos=pos, y_pos=pos))
    torch.testing.assert_close(rows_xy, rows_sc, rtol=2e-5, atol=0)          # scale invariance
    assert float(run_rows(w1, x, x, dict(x_pos=pos, y_pos=pos)).abs().max()) == 0.0   # identity
    k = 5                                                                    # shift by k bins -> k/(N-1)
    xs = torch.zeros(4, 2048, device=device())
    xs[:, 100:600] = x[:4, 100:600]
    ys = torch.roll(xs, k, dims=1)
    got = run_rows(w1, xs, ys, dict(x_pos=pos, y_pos=pos))
    torch.testing.assert_close(got, torch.full_like(got, k / 2047.0), rtol=3e-4, atol=0)  # analytic value, fp32 conditioning
    from oracle import sot_oracle as so
    want = so.forward(xs.cpu().numpy(), ys.cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(), p=1.0, flags=so.make_flags())
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL)


def test_known_answers_and_errors():
    from sot_amd.losses import Wasserstein1D, wasserstein_1d
    native()
    dev = device()
    n = 64
    pos = torch.linspace(0, 1, n, device=dev)
    a = torch.zeros(1, n, device=dev)
    b = torch.zeros(1, n, device=dev)
    a[0, 10] = 3.0
    b[0, 37] = 0.5
    d = (pos[10] - pos[37]).abs()
    assert float(Wasserstein1D(p=1)(a, b, x_pos=pos, y_pos=pos)) == float(d)
    assert float(Wasserstein1D(p=2)(a, b, x_pos=pos, y_pos=pos)) == float(d * d)
    with pytest.raises(ValueError):
        Wasserstein1D(p=1)(a, b)
    with pytest.raises(AssertionError):
        Wasserstein1D(p=0.5)(a, b, x_pos=pos, y_pos=pos)
    with pytest.raises(AssertionError):
        wasserstein_1d(pos[None], pos[None], a, b, p=0.5)
    # CPU tensors take the package's torch-op route (tests/test_cpu_path.py) and give the same known answer
    assert float(Wasserstein1D(p=1)(a.cpu(), b.cpu(), x_pos=pos.cpu(), y_pos=pos.cpu())) == float(d)
    # fixed_x buffer form (metrics.py:148) under inference_mode
    with torch.inference_mode():
        m = Wasserstein1D(p=2, fixed_x=n).to(dev)
        assert float(m(a, b)) == float(d * d)
    # functional form: weights used as given, default uniform weights
    w = wasserstein_1d(pos[None].expand(2, n), pos[None].expand(2, n) * 0.5)
    exp = (pos - 0.5 * pos).abs().mean()
    torch.testing.assert_close(w, exp.expand(2), rtol=1e-5, atol=0)


def test_host_semantics_dims_hinge_strides():
    from oracle import torch_restatement as tr
    from oracle.inputs import gen_inputs
    from sot_amd.losses import MixOfLosses, Wasserstein1D
    native()
    dev = device()
    x, y = gen_inputs("peaky", 12, 40, 40, 21)
    pos = torch.linspace(0, 1, 40)
    x3, y3 = x.reshape(3, 4, 40), y.reshape(3, 4, 40)
    mod = Wasserstein1D(p=2, square_dist=True, hinge=True).to(dev)
    for dims in (None, [1], [0], [0, 1]):
        got = mod(x3.to(dev), y3.to(dev), x_pos=pos.to(dev), y_pos=pos.to(dev), dims=dims, hinge=0.003)
        want = tr.sot_loss(x3, y3, pos, pos, p=2, square_dist=True, hinge=True, hinge_value=0.003, dims=dims)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-9)
    # call-time kwargs are OR-ed with ctor flags (losses.py:180,194-195)
    m2 = Wasserstein1D(p=2, square_dist=True).to(dev)
    got = m2(x.to(dev), y.to(dev), x_pos=pos.to(dev), y_pos=pos.to(dev), dont_normalize=True, limit_quantile_range=True)
    want = tr.sot_loss(x, y, pos, pos, p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=0)
    # non-contiguous inputs (a strided view) are accepted
    wide = torch.zeros(12, 80)
    wide[:, ::2] = x
    got = m2(wide.to(dev)[:, ::2], y.to(dev), x_pos=pos.to(dev), y_pos=pos.to(dev))
    want = tr.sot_loss(x, y, pos, pos, p=2, square_dist=True)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=0)
    # row-strided view (rows of a wider matrix)
    wide2 = torch.rand(12, 100)
    wide2[:, :40] = x
    got = m2(wide2.to(dev)[:, :40], y.to(dev), x_pos=pos.to(dev), y_pos=pos.to(dev))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=0)
    # MixOfLosses passthrough (losses.py:346-362; trainer.py:220)
    mix = MixOfLosses([m2], [0.5])
    out = mix(x.to(dev), y.to(dev), x_pos=pos.to(dev), y_pos=pos.to(dev))
    assert list(out) == ["Wasserstein1D"]
    torch.testing.assert_close(out["Wasserstein1D"].cpu(), 0.5 * want, rtol=1e-5, atol=0)


@pytest.mark.parametrize("n", [1, 2, 5, 64, 100, 257, 1000, 1025, 2048, 3000, 5000])
def test_segmented_sort_bit_exact_indices(n):
    nat = native()
    g = torch.Generator().manual_seed(n)
    # distinct keys (a shuffled grid plus a tiny jitter): the permutation is unique, so torch's unstable sort agrees
    keys = torch.stack([(torch.randperm(n, generator=g).float() + 0.25 * torch.rand(n, generator=g)) / n for _ in range(7)])
    want_v, want_i = torch.sort(keys, 1)
    got_v, got_i = nat.segmented_sort(keys.to(device()))
    assert torch.equal(got_v.cpu(), want_v)
    assert torch.equal(got_i.cpu(), want_i)
    # ties: stable (lowest index first)
    tied = torch.randint(0, 4, (3, n), generator=g).float()
    want_v, want_i = torch.sort(tied, dim=1, stable=True)
    got_v, got_i = nat.segmented_sort(tied.to(device()))
    assert torch.equal(got_v.cpu(), want_v) and torch.equal(got_i.cpu(), want_i)


@pytest.mark.parametrize("n", [2, 63, 64, 65, 129, 257, 500, 512, 513, 1024, 1025, 1500, 2047, 2048])
def test_segmented_sort_adversarial_keys(n):
    """Round 6: rows of <= 2048 keys are sorted by ONE wavefront on one packed word per key (quantised key | index), exact by a repair of
    the runs that share a quantisation bin, with the in-LDS merge sort as the fallback (csrc/sot_wave_sort.hpp).  Values AND indices
    bit-identical to torch.sort(stable=True) on the cases that exercise each branch: the adaptive range (all keys in one 2^-14 interval),
    the repair (near-ties), the fallback (everything but one key in ONE bin; all keys equal; +-inf; an overflowing range; a denormal
    range), -0 == +0, and NaN rows beside clean rows (NaN has no defined place; indices stay inside the row, clean rows stay exact)."""
    nat = native()
    dev = device()
    g = torch.Generator().manual_seed(1000 + n)
    B = 19
    u = torch.rand(B, n, generator=g)

    def same(keys):
        got_v, got_i = nat.segmented_sort(keys.to(dev))
        want_v, want_i = torch.sort(keys, dim=1, stable=True)
        assert torch.equal(got_v.cpu(), want_v) and torch.equal(got_i.cpu(), want_i)   # (value equality: the merge-sort fallback returns +0 for -0)

    same(u)
    same(torch.randn(B, n, generator=g) * 3)
    same(torch.round(u * 50) / 50)                                         # ties: long runs of equal keys (declined: merge sort) and short ones
    same(torch.sort(u, dim=1, descending=True)[0].contiguous())
    same(0.5 + u * 2.0 ** -14)                                             # one bin of the float's own top bits; separated by the row's range
    near = u.clone(); near[:, 1::2] = torch.nextafter(near[:, 0::2][:, :near[:, 1::2].shape[1]], torch.tensor(2.0)); same(near)   # pairs one ulp apart
    same(torch.cat([u[:, :-1] * 1e-9, torch.ones(B, 1)], 1))               # everything but one key in ONE bin: the fallback
    if n > 100:
        mod = u.clone(); mod[:, : n - n // 40] *= 0.01; same(mod)           # moderately clustered: the distribution form's buckets overflow, the network takes the row
    same(torch.full((B, n), 0.25))
    z = u.clone(); z[:, ::3] = 0.0; z[:, 1::5] = -0.0; same(z)
    for bad in (float("inf"), float("-inf")):
        w = u.clone(); w[:, n // 2] = bad; same(w)
    same((u - 0.5) * 6e38)                                                 # max - min overflows
    same(u * 1e-42)                                                        # a denormal range: the scale overflows
    if n > 2:
        w = u.clone(); w[::2, 1] = float("nan")
        got_v, got_i = nat.segmented_sort(w.to(dev))
        assert int(got_i.min()) >= 0 and int(got_i.max()) < n
        want_v, want_i = torch.sort(w[1::2], dim=1, stable=True)
        assert torch.equal(got_v.cpu()[1::2], want_v) and torch.equal(got_i.cpu()[1::2], want_i)
    # strided rows and a base pointer off the 16-byte grid (the scalar-load variant)
    wide = torch.rand(B, n + 5, generator=g).to(dev)
    got_v, got_i = nat.segmented_sort(wide[:, 1:n + 1])
    want_v, want_i = torch.sort(wide[:, 1:n + 1].cpu(), dim=1, stable=True)
    assert torch.equal(got_v.cpu(), want_v) and torch.equal(got_i.cpu(), want_i)


@pytest.mark.parametrize("shape", [(9, 2, 2), (5, 50, 70), (7, 300, 411), (6, 512, 512), (5, 1000, 1024), (7, 1024, 1024), (4, 1025, 1025), (4, 1536, 1400), (6, 2048, 2048),
                                   (5, 2048, 2000)])
@pytest.mark.parametrize("mode", ["p1", "cutoff", "nocut", "p3"])
def test_rowpos_presort_kernel_on_every_route(shape, mode):
    """Round 6: per-row positions that nobody has sorted are sorted AHEAD of the row kernel by sot_rowpos_sort_kernel (one wavefront per row,
    permutations into the caller's row_perm_out or the workspace); the row kernels gather through them.  Arrays that arrive sorted get the
    identity; arrays the wave sort declines (clustered / tied / non-finite positions) are merge-sorted by the same wavefront: the image is always
    complete.  Every route must give the same bits: default (pre-sort) == SOT_FLAG_NO_SPECIALIZE (the row kernel's own merge sort) == the
    oracle within the forward tolerance; the permutations left in row_perm_out are the stable argsort on EVERY row."""
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    from oracle.make_golden import MODES
    nat = native()
    dev = device()
    B, n, m = shape
    x, y = gen_inputs("peaky", B, n, m, 300 + n + m)
    g = torch.Generator().manual_seed(n * 3 + m)
    xp, yp = torch.rand(B, n, generator=g), torch.rand(B, m, generator=g)
    xp[0], yp[0] = torch.sort(xp[0]).values, torch.sort(yp[0]).values          # both sorted: identities
    xp[1] = torch.sort(xp[1]).values                                            # one sorted, one not: identity + permutation
    if n > 12:
        xp[2, : n - 1] *= 1e-9; xp[2, n - 1] = 1.0                              # clustered: the wave sort declines (merge sort by the same wavefront)
        yp[3] = torch.round(yp[3] * 3) / 3                                      # long runs of ties: declined
    p, flags = ctor_to_flags(MODES[mode])
    xd, yd, xpd, ypd = x.to(dev), y.to(dev), xp.to(dev), yp.to(dev)
    want = so.forward(x.numpy(), y.numpy(), xp.numpy(), yp.numpy(), p=p, flags=flags)
    rows_pre = nat.forward_rows(xd, yd, xpd, ypd, p, flags, None)                         # pre-sort into the workspace
    rows_own = nat.forward_rows(xd, yd, xpd, ypd, p, flags | nat.FLAG_NO_SPECIALIZE, None)  # the row kernel's own sort
    perm = nat.row_permutations(xd, yd, xpd, ypd, flags)
    perm.fill_(12345 % 65536)
    rows_out = nat.forward_rows(xd, yd, xpd, ypd, p, flags, None, perm_out=perm)           # pre-sort into the caller's image
    rows_in = nat.forward_rows(xd, yd, xpd, ypd, p, flags, None, perm_in=perm)
    assert torch.equal(rows_pre, rows_own) and torch.equal(rows_out, rows_own) and torch.equal(rows_in, rows_own)
    np.testing.assert_allclose(rows_pre.cpu().numpy(), want, rtol=RTOL)
    got = perm.cpu().to(torch.int64)
    assert torch.equal(got[:, :n], torch.sort(xp, dim=1, stable=True).indices) and torch.equal(got[:, n:], torch.sort(yp, dim=1, stable=True).indices)
    one = torch.ones(1, device=dev)
    gx0, gy0 = nat.backward_rows(xd, yd, xpd, ypd, p, flags | nat.FLAG_NO_SPECIALIZE, one)
    gx1, gy1 = nat.backward_rows(xd, yd, xpd, ypd, p, flags, one)                          # pre-sort into ITS workspace
    gx2, gy2 = nat.backward_rows(xd, yd, xpd, ypd, p, flags, one, perm_in=perm)
    assert torch.equal(gx0, gx1) and torch.equal(gy0, gy1) and torch.equal(gx0, gx2) and torch.equal(gy0, gy2)
    px0 = nat.position_grads(xd, yd, xpd, ypd, p, flags | nat.FLAG_NO_SPECIALIZE, one)
    px1 = nat.position_grads(xd, yd, xpd, ypd, p, flags, one)
    assert torch.equal(px0[0], px1[0]) and torch.equal(px0[1], px1[1])


@pytest.mark.parametrize("shape", [(5, 50, 70), (3, 300, 300), (9, 1025, 1025), (2, 2048, 2048)])
@pytest.mark.parametrize("mode", ["p1", "cutoff"])
def test_unsorted_positions_shared_and_per_row(shape, mode):
    """require_sort does real work: shared unsorted grid (plan path) and per-row grids (in-LDS sort)."""
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    from oracle.make_golden import MODES
    B, n, m = shape
    x, y = gen_inputs("uniform", B, n, m, 5 + n)
    g = torch.Generator().manual_seed(n)
    p, flags = ctor_to_flags(MODES[mode])
    mod = module_for(MODES[mode])
    for per_row in (False, True):
        xp = torch.rand((B, n) if per_row else (n,), generator=g)
        yp = torch.rand((B, m) if per_row else (m,), generator=g)
        want = so.forward(x.numpy(), y.numpy(), xp.numpy(), yp.numpy(), p=p, flags=flags)
        xd = x.to(device()).requires_grad_(True)
        yd = y.to(device()).requires_grad_(True)
        rows = run_rows(mod, xd, yd, dict(x_pos=xp.to(device()), y_pos=yp.to(device())))
        np.testing.assert_allclose(rows.detach().cpu().numpy(), want, rtol=RTOL)
        rows.sum().backward()
        gx, gy = so.backward(x.numpy(), y.numpy(), xp.numpy(), yp.numpy(), np.ones(B, np.float32), p=p, flags=flags)
        for got, ref in ((xd.grad, gx), (yd.grad, gy)):
            tol = 1e-5 * np.abs(ref).max(axis=1, keepdims=True) + 3e-8
            assert (np.abs(got.cpu().numpy() - ref) <= tol).all()


@pytest.mark.parametrize("shape", [(5, 50, 70), (7, 300, 411), (9, 1025, 1025), (6, 2048, 2048), (3, 4096, 3000), (70, 129, 129)])
@pytest.mark.parametrize("mode", ["p1", "cutoff"])
def test_per_row_supports_are_sorted_once_per_step(shape, mode):
    """Round 5: per-row supports (`torch.sort(u_values, 1)` on every row, losses.py:286-288) are sorted by the forward only, which leaves each
    row's two permutations in a uint16 buffer (sot_problem.row_perm_out); the backward and the position-gradient kernels gather through them
    (row_perm_in).  The permutations are the STABLE argsort of the positions (distinct keys: torch's own order); gradients with the hand-over
    are bit-identical to the kernels sorting on their own, to the module's autograd (which uses the hand-over), and a forward that is handed
    permutations returns the same rows; rows that arrive sorted give the identity."""
    from oracle.inputs import gen_inputs
    from oracle.make_golden import MODES
    nat = native()
    dev = device()
    B, n, m = shape
    x, y = gen_inputs("peaky", B, n, m, 77 + n)
    g = torch.Generator().manual_seed(n + m)
    xp, yp = torch.rand(B, n, generator=g), torch.rand(B, m, generator=g)
    xp[1] = torch.sort(xp[1]).values                     # one row arrives sorted
    p, flags = ctor_to_flags(MODES[mode])
    xd, yd, xpd, ypd = x.to(dev), y.to(dev), xp.to(dev), yp.to(dev)
    perm = nat.row_permutations(xd, yd, xpd, ypd, flags)
    assert perm is not None and perm.shape == (B, n + m) and perm.dtype == torch.uint16
    rows = nat.forward_rows(xd, yd, xpd, ypd, p, flags, None, perm_out=perm)
    assert torch.equal(rows, nat.forward_rows(xd, yd, xpd, ypd, p, flags, None))
    assert torch.equal(rows, nat.forward_rows(xd, yd, xpd, ypd, p, flags, None, perm_in=perm))
    pc = perm.cpu().numpy().astype(np.int64)
    assert np.array_equal(pc[:, :n], np.argsort(xp.numpy(), axis=1, kind="stable"))
    assert np.array_equal(pc[:, n:], np.argsort(yp.numpy(), axis=1, kind="stable"))
    assert np.array_equal(pc[1, :n], np.arange(n))
    up = torch.rand(B, generator=g).to(dev)
    gx0, gy0 = nat.backward_rows(xd, yd, xpd, ypd, p, flags, up)
    gx1, gy1 = nat.backward_rows(xd, yd, xpd, ypd, p, flags, up, perm_in=perm)
    assert torch.equal(gx0, gx1) and torch.equal(gy0, gy1)
    px0, py0 = nat.position_grads(xd, yd, xpd, ypd, p, flags, up)
    px1, py1 = nat.position_grads(xd, yd, xpd, ypd, p, flags, up, perm_in=perm)
    assert torch.equal(px0, px1) and torch.equal(py0, py1)
    # the module's autograd node
    mod = module_for(MODES[mode])
    ts = [t.clone().requires_grad_(True) for t in (xd, yd, xpd, ypd)]
    out = run_rows(mod, ts[0], ts[1], dict(x_pos=ts[2], y_pos=ts[3]))
    (out * up).sum().backward()
    assert torch.equal(out.detach(), rows)
    for got, want in zip((t.grad for t in ts), (gx0, gy0, px0, py0)):
        assert torch.equal(got, want)


@pytest.mark.parametrize("shape", [(3, 4000, 4000), (2, 8192, 8192), (2, 9000, 700), (300, 33, 2049)])
def test_large_and_ragged_sizes_against_oracle(shape):
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    B, n, m = shape
    x, y = gen_inputs("peaky", B, n, m, 1000 + n)
    xp, yp = torch.linspace(0, 1, n), torch.linspace(0, 1, m)
    for ctor in (dict(p=1), dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)):
        p, flags = ctor_to_flags(ctor)
        want = so.forward(x.numpy(), y.numpy(), xp.numpy(), yp.numpy(), p=p, flags=flags)
        rows = run_rows(module_for(ctor), x.to(device()), y.to(device()), dict(x_pos=xp.to(device()), y_pos=yp.to(device())))
        np.testing.assert_allclose(rows.cpu().numpy(), want, rtol=RTOL, atol=1e-12)


def test_size_limit_is_an_error_not_a_crash():
    from sot_amd import _native as nat
    native()
    dev = device()
    g = torch.Generator(device=dev).manual_seed(30000)
    pos = torch.linspace(0, 1, 30000, device=dev)
    x = torch.rand(2, 30000, device=dev, generator=g) * (1.0 - pos)     # two clearly different measures: a loss of O(0.1), so that the
    y = torch.rand(2, 30000, device=dev, generator=g) * pos             # two ATen backends' summation orders stay a relative 1e-5 effect
    from sot_amd.losses import Wasserstein1D
    with pytest.raises(nat.SotError) as info:          # the NATIVE layer refuses rows that do not fit one CU's LDS ...
        nat.forward_rows(x, y, pos, pos.clone(), 1.0, 8)
    assert info.value.status == nat.SOT_ERR_UNSUPPORTED_SIZE
    # ... and the module, like the reference (no size limit in losses.py:223-313), still answers: torch ops on the GPU, said once
    got = Wasserstein1D(p=1)(x, y, x_pos=pos, y_pos=pos)
    want = Wasserstein1D(p=1)(x.cpu(), y.cpu(), x_pos=pos.cpu(), y_pos=pos.cpu())
    # (ATen's GPU cumsum / sum associate differently from its CPU ones: 30000-point rows of near-identical CDFs agree to ~2e-5)
    assert got.is_cuda and abs(float(got) - float(want)) <= 1e-4 * abs(float(want))


def test_masked_dense_equals_ragged_removal():
    """BASELINE config 4 semantics: zero-weight points are inert, i.e. masking == removing support points."""
    from oracle.inputs import gen_inputs
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    B, N = 64, 512
    x, y = gen_inputs("peaky", B, N, N, 1234)
    g = torch.Generator().manual_seed(1234)
    tau = 10 ** (-3 + 2.7 * torch.rand(B, 1, generator=g))
    xm = torch.where(x < tau * x.amax(1, keepdim=True), torch.zeros_like(x), x)
    ym = torch.where(y < tau * y.amax(1, keepdim=True), torch.zeros_like(y), y)
    pos = torch.linspace(0, 1, N)
    mod = Wasserstein1D(p=1).to(dev)
    dense = run_rows(mod, xm.to(dev), ym.to(dev), dict(x_pos=pos.to(dev), y_pos=pos.to(dev))).cpu()
    for r in range(0, B, 7):  # truly ragged evaluation of a few rows: only the kept support points
        kx, ky = xm[r] > 0, ym[r] > 0
        one = mod(xm[r][kx][None].to(dev), ym[r][ky][None].to(dev), x_pos=pos[kx].to(dev), y_pos=pos[ky].to(dev)).cpu()
        torch.testing.assert_close(one, dense[r], rtol=5e-6, atol=0)


@pytest.mark.parametrize("m", [1, 8])
def test_row_constant_division_is_ieee_exact(m):
    """The kernel divides by the row mass with a reciprocal + FMA-residual correction (IEEE fallback for tiny
    operands).  Pin it against IEEE fp32 division on ~2M quotients: x row = [S] (n = 1, so the ATen-order
    mass is S itself), dont_normalize => b_0 = y_0 / S and V_0 = fl32((double) b_0) = b_0."""
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    rng = np.random.default_rng(m)
    B = 1 << 20
    # divisors: random mantissas over many exponents + adversarial patterns (all-ones / one-bit mantissas)
    mant = rng.integers(0, 1 << 23, B, dtype=np.uint32)
    mant[: B // 8] = (1 << 23) - 1 - rng.integers(0, 4, B // 8, dtype=np.uint32)
    mant[B // 8: B // 4] = rng.integers(0, 4, B // 8, dtype=np.uint32)
    expo = rng.integers(127 - 20, 127 + 30, B, dtype=np.uint32)
    S = ((expo << 23) | mant).view(np.float32)
    # numerators: random mantissas, exponents from deep subnormal quotients up to > S (dont_normalize allows b > 1)
    ymant = rng.integers(0, 1 << 23, (B, m), dtype=np.uint32)
    yexp = rng.integers(1, 127 + 35, (B, m), dtype=np.uint32)
    y = ((yexp << 23) | ymant).view(np.float32)
    y[rng.random((B, m)) < 0.05] = 0.0
    y[rng.random((B, m)) < 0.02] *= np.float32(1e-30)  # subnormal numerators / quotients
    want = (y[:, 0] / S).astype(np.float32)            # numpy: IEEE fp32 division
    mod = Wasserstein1D(p=1, dont_normalize=True, require_sort=False).to(dev)
    px = torch.zeros(1, device=dev)
    py = torch.linspace(0, 1, m, device=dev)
    out = mod(to_dev(S[:, None]), to_dev(y), x_pos=px, y_pos=py, return_quantiles=True)
    V0 = out[4][:, 0].cpu().numpy()
    bad = V0.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), (int(bad.sum()), S[bad][:4], y[bad, 0][:4], V0[bad][:4], want[bad][:4])


def test_fused_mean_matches_two_kernel_path_and_is_repeatable():
    from oracle.inputs import gen_inputs
    nat = native()
    dev = device()
    for (B, N) in ((4, 512), (1000, 1025), (8192, 512), (3000, 2048)):
        x, y = gen_inputs("peaky", B, N, N, B)
        x, y = x.to(dev), y.to(dev)
        pos = torch.linspace(0, 1, N, device=dev)
        rows = nat.forward_rows(x, y, pos, pos, 1.0, nat.FLAG_REQUIRE_SORT)
        want = nat.reduce_mean(rows)
        got = [nat.loss_fused(x, y, pos, pos, 1.0, nat.FLAG_REQUIRE_SORT)[0] for _ in range(20)]
        torch.cuda.synchronize()
        assert all(torch.equal(g, got[0]) for g in got)
        assert torch.equal(got[0], want)  # same reduce kernel behind both entry points
        assert torch.equal(nat.loss_fused(x, y, pos, pos, 1.0, nat.FLAG_REQUIRE_SORT)[1], rows)
        # hinge folded into the fused reduction (losses.py:203-205)
        h = float(rows.median())
        got_h = nat.loss_fused(x, y, pos, pos, 1.0, nat.FLAG_REQUIRE_SORT, hinge=h)[0]
        torch.testing.assert_close(got_h, torch.relu(rows - h).double().mean().float(), rtol=1e-6, atol=0)


def test_streams_and_graph_capture():
    """The launch path neither synchronises nor allocates outside torch's allocator: it can run on side
    streams concurrently and be captured into a HIP graph and replayed."""
    from oracle.inputs import gen_inputs
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    B, N = 2048, 1025
    x, y = gen_inputs("peaky", B, N, N, 9)
    x, y = x.to(dev), y.to(dev)
    pos = torch.linspace(0, 1, N, device=dev)
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
    with torch.no_grad():
        # first use of a fresh module happens on a SIDE stream: the plan is created there and the default stream's
        # call right after must wait for it (plan.ready event), without any host synchronisation in between
        fresh = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            first = fresh(x, y, x_pos=pos, y_pos=pos)
        second = fresh(x, y, x_pos=pos, y_pos=pos)
        torch.cuda.synchronize()
        assert torch.equal(first, second)
        want = mod(x, y, x_pos=pos, y_pos=pos).clone()
        assert torch.equal(first, want)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for s in (s1, s2, s1, s2):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                outs.append(mod(x, y, x_pos=pos, y_pos=pos))
        torch.cuda.synchronize()
        assert all(torch.equal(o, want) for o in outs)
        # graph capture + replay on fresh data written into the static input buffers
        sx, sy = x.clone(), y.clone()
        mod(sx, sy, x_pos=pos, y_pos=pos)  # warm: plan + ticket exist before capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = mod(sx, sy, x_pos=pos, y_pos=pos)
        x2, y2 = gen_inputs("uniform", B, N, N, 10)
        sx.copy_(x2.to(dev))
        sy.copy_(y2.to(dev))
        g.replay()
        torch.cuda.synchronize()
        ref = mod(x2.to(dev), y2.to(dev), x_pos=pos, y_pos=pos)
        assert torch.equal(out, ref)


def _to_csr(dense, pos, keep):
    """concatenate kept entries row by row -> (weights, positions, offsets[int64])"""
    lens = keep.sum(1)
    off = torch.zeros(dense.shape[0] + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    posb = pos.expand_as(dense) if pos.ndim == 1 else pos
    return dense[keep], posb[keep], off


@pytest.mark.parametrize("mode", ["p1", "cutoff", "nocut"])
@pytest.mark.parametrize("N,B", [(512, 512), (257, 300), (2048, 64)])
def test_csr_ragged_matches_masked_dense_and_oracle(mode, N, B):
    """BASELINE config 4: per-row amplitude cutoff tau_r = 10^U[-3,-0.3] * max_r on peaky spectra -> ragged supports.
    The CSR kernel must agree with (a) the dense kernel on the zero-masked rows and (b) the oracle evaluated on the
    truly ragged rows (only the kept support points)."""
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    from oracle.make_golden import MODES
    from sot_amd.losses import wasserstein_1d_csr
    native()
    dev = device()
    x, y = gen_inputs("peaky", B, N, N, 1234)
    g = torch.Generator().manual_seed(1234)
    tau = 10 ** (-3 + 2.7 * torch.rand(B, 1, generator=g))
    kx, ky = x >= tau * x.amax(1, keepdim=True), y >= tau * y.amax(1, keepdim=True)
    xm, ym = torch.where(kx, x, torch.zeros_like(x)), torch.where(ky, y, torch.zeros_like(y))
    pos = torch.linspace(0, 1, N)
    ctor = MODES[mode]
    p, flags = ctor_to_flags(ctor)
    xw, xp, xo = _to_csr(x, pos, kx)
    yw, yp, yo = _to_csr(y, pos, ky)
    max_n, max_m = int(kx.sum(1).max()), int(ky.sum(1).max())
    kw = dict(p=ctor.get("p", 1), square_dist=ctor.get("square_dist", False), dont_normalize=ctor.get("dont_normalize", False),
              limit_quantile_range=ctor.get("limit_quantile_range", False))
    got = wasserstein_1d_csr(xw.to(dev), xp.to(dev), xo.to(dev), yw.to(dev), yp.to(dev), yo.to(dev), max_n, max_m, **kw).cpu().numpy()
    dense = run_rows(module_for(ctor), xm.to(dev), ym.to(dev), dict(x_pos=pos.to(dev), y_pos=pos.to(dev))).cpu().numpy()
    # masking == removing, up to the different summation grouping of the (identical) non-zero weights in the row
    # mass; with the cutoff a 1-ulp mass difference is amplified by the Q_k > 1 knife-edge, so there
    # only the bulk of the rows is compared against the dense form and the oracle check below is the exact one
    if ctor.get("limit_quantile_range", False):
        assert (np.abs(got - dense) <= 2e-5 * np.abs(dense) + 1e-10).mean() > 0.5  # ~40 % of rows sit on the knife-edge
    else:
        np.testing.assert_allclose(got, dense, rtol=2e-5, atol=1e-10)
    for r in range(0, B, max(1, B // 16)):  # oracle on the truly ragged row: bit-level agreement expected (same S order)
        want = so.forward(x[r][kx[r]][None].numpy(), y[r][ky[r]][None].numpy(), pos[kx[r]].numpy(), pos[ky[r]].numpy(), p=p, flags=flags)
        np.testing.assert_allclose(got[r], want[0], rtol=RTOL, atol=1e-12)


def test_csr_unsorted_rows_and_invalid_rows():
    from oracle import sot_oracle as so
    from sot_amd.losses import wasserstein_1d_csr
    native()
    dev = device()
    g = torch.Generator().manual_seed(5)
    lens_x = [7, 1, 300, 64, 0, 129]   # row 4 is empty -> NaN
    lens_y = [9, 5, 280, 64, 3, 400]   # row 5 exceeds max_m below -> NaN
    xw = [torch.rand(l, generator=g) for l in lens_x]
    yw = [torch.rand(l, generator=g) for l in lens_y]
    xp = [torch.rand(l, generator=g) for l in lens_x]  # unsorted positions: the per-row LDS sort has to run
    yp = [torch.rand(l, generator=g) for l in lens_y]
    off = lambda ls: torch.tensor([0] + list(np.cumsum(ls)), dtype=torch.int64)
    got = wasserstein_1d_csr(torch.cat(xw).to(dev), torch.cat(xp).to(dev), off(lens_x).to(dev), torch.cat(yw).to(dev),
                             torch.cat(yp).to(dev), off(lens_y).to(dev), 300, 300, p=1).cpu().numpy()
    assert np.isnan(got[4]) and np.isnan(got[5])
    for r in (0, 1, 2, 3):
        want = so.forward(xw[r][None].numpy(), yw[r][None].numpy(), xp[r].numpy(), yp[r].numpy(), p=1.0, flags=so.make_flags())
        np.testing.assert_allclose(got[r], want[0], rtol=RTOL)


def test_training_step_slice_config5(manifest):
    """BASELINE config 5: harmonic clips -> flattop STFT magnitudes (n_fft 2048, hop 256, 16 frames) -> SOT paper-cutoff
    forward + backward.  The producer (torch.stft on the GPU) must reproduce the spectra the REFERENCE's TorchSTFT
    produced for the same audio (fixture), and loss / gradient w.r.t. the estimate's spectrum must match the oracle."""
    from oracle import sot_oracle as so
    from oracle.inputs import harmonic_audio_pair
    from oracle.make_golden import MODES
    from sot_amd import spectra
    native()
    dev = device()
    fx = dict(np.load(os.path.join(GOLDEN, "inputs_harmonic_stft.npz")))
    ax, ay = harmonic_audio_pair(nb=2, seed=11)
    sx = spectra.stft_magnitude(ax.to(dev))
    sy = spectra.stft_magnitude(ay.to(dev))
    assert tuple(sx.shape) == fx["x"].shape == (2, 16, 1025) and sx.is_contiguous()
    for got, want in ((sx, fx["x"]), (sy, fx["y"])):
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max()
    pos = spectra.unit_frequencies(2048, 16000.0, dev)
    np.testing.assert_allclose(pos.cpu().numpy(), fx["x_pos"], rtol=0, atol=0)
    # end to end: gradient reaches the estimate's AUDIO through torch.stft's autograd and our backward kernel
    ay_d = ay.to(dev).requires_grad_(True)
    mod = module_for(MODES["cutoff"])
    loss = spectra.training_step_slice(mod, ax.to(dev), ay_d)
    loss.backward()
    assert torch.isfinite(ay_d.grad).all() and float(ay_d.grad.abs().max()) > 0
    # loss on the fixture spectra == the reference's scalar for those spectra
    ref = manifest["harmonic_stft_cutoff"]["scalar"]
    got = float(mod(to_dev(fx["x"]), to_dev(fx["y"]), x_pos=pos, y_pos=pos.clone()))
    assert abs(got - ref) <= RTOL * abs(ref)
    # and the generator produces well-formed clips
    g = torch.Generator(device=dev).manual_seed(3)
    clips = spectra.harmonic_batch(16, generator=g, device=dev)
    assert clips.shape == (16, 4096) and abs(float(clips.abs().amax(1).mean()) - 0.9) < 1e-5


def test_sharded_loss_single_rank_equals_module():
    """sot_amd.distributed.sharded_sot_loss with one rank (no process group): value and gradient equal the module's;
    the N>1 reduction logic itself is covered by the gloo world-size-2 CPU test and by bench.py's RCCL path."""
    from oracle.inputs import gen_inputs
    from sot_amd.distributed import sharded_sot_loss
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    x, y = gen_inputs("peaky", 300, 257, 257, 77)
    pos = torch.linspace(0, 1, 257, device=dev)
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
    y1 = y.to(dev).requires_grad_(True)
    y2 = y.to(dev).requires_grad_(True)
    a = mod(x.to(dev), y1, x_pos=pos, y_pos=pos)
    b = sharded_sot_loss(mod, x.to(dev), y2, x_pos=pos, y_pos=pos)
    a.backward()
    b.backward()
    torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
    torch.testing.assert_close(y1.grad, y2.grad, rtol=1e-5, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("N,B", [(2048, 70), (1025, 33), (257, 130), (512, 9), (300, 17)])
@pytest.mark.parametrize("flags,p", [(8, 1.0), (1 | 2 | 4 | 8, 2.0), (4 | 8, 2.0)])
def test_loss_and_grad_in_one_pass_matches_forward_plus_backward(N, B, flags, p):
    """sot_w1d_loss_and_grad (the training form: the y-only backward kernel also accumulates the row losses) against the
    separate forward and backward calls: row losses, mean and gradient bit for bit (N = 300 has no compile-time kernel and
    takes the three-kernel fallback inside the library)."""
    from oracle.inputs import gen_inputs
    nat = native()
    x, y = gen_inputs("peaky", B, N, N, 99 + N)
    x, y = x.to(device()), y.to(device())
    pos = torch.linspace(0, 1, N).to(device())
    pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2)
    mean, rows, gy = nat.loss_and_grad(x, y, pos, pos2, p, flags, plan)
    # (p = 1 on one grid: the forward alone would take the merge-free kernel, whose rows agree to 2e-6, not bit for bit)
    mean2, rows2, _ = nat.loss_fused(x, y, pos, pos2, p, flags | nat.FLAG_NO_AREA, plan)
    rows3 = nat.forward_rows(x, y, pos, pos2, p, flags, plan)
    torch.testing.assert_close(rows3, rows, rtol=2e-6, atol=1e-12)
    one = torch.ones((), device=device())
    _, gy2 = nat.backward_rows(x, y, pos, pos2, p, flags, one, need_gx=False, plan=plan, grad_scale=1.0 / B)
    assert torch.equal(rows, rows2), float((rows - rows2).abs().max())
    assert torch.equal(mean, mean2)
    assert torch.equal(gy, gy2), float((gy - gy2).abs().max())
    # the rescaling kernel: exactly nothing for 1, a plain product otherwise
    keep = gy.clone()
    assert torch.equal(nat.scale_inplace(gy, one), keep)
    three = torch.full((), 3.0, device=device())
    assert torch.equal(nat.scale_inplace(gy, three), keep * 3.0)


@pytest.mark.gpu
def test_module_early_gradient_matches_the_two_pass_form():
    """Wasserstein1D with y.requires_grad only: loss and gradient are identical whether the gradient is computed with the
    loss (default) or in backward; upstream scaling and a second backward through a retained graph work."""
    from oracle.inputs import gen_inputs
    from oracle.make_golden import MODES
    from sot_amd import losses
    x, y = gen_inputs("peaky", 64, 1025, 1025, 5)
    x = x.to(device())
    pos = torch.linspace(0, 1, 1025).to(device())
    mod = module_for(MODES["cutoff"])
    results = []
    for early in (True, False):
        losses.EARLY_GRADIENT = early
        try:
            yv = y.to(device()).requires_grad_(True)
            loss = mod(x, yv, x_pos=pos, y_pos=pos.clone())
            (2.5 * loss).backward(retain_graph=True)
            g1 = yv.grad.clone()
            yv.grad = None
            loss.backward()
            results.append((loss.detach().clone(), g1, yv.grad.clone()))
        finally:
            losses.EARLY_GRADIENT = True
    (l1, a1, b1), (l2, a2, b2) = results
    assert torch.equal(l1, l2)
    assert torch.equal(b1, b2)                      # second backward: recomputed by the backward kernel in both forms
    torch.testing.assert_close(a1, a2, rtol=2e-7, atol=1e-37)   # 2.5 * (g / B) rounded in one or two steps (subnormals: atol)
    torch.testing.assert_close(a1, 2.5 * b1, rtol=2e-7, atol=1e-37)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(1, 2048), (7, 300), (130, 257), (1000, 1025), (8192, 2048), (65536, 257), (20000, 512)])
def test_in_kernel_batch_mean_is_bit_identical_to_the_mean_kernel(B, N):
    """sot_w1d_loss / sot_w1d_loss_and_grad with completion counters: the last workgroup of the row kernel reduces the row
    losses (losses.py:203-211) in the order of sot_w1d_reduce_mean -- same bits, every time, also with the hinge, from two
    streams, and the counters are left zero (any later launch works)."""
    nat = native()
    dev = device()
    g = torch.Generator(device=dev).manual_seed(B + N)
    x, y = torch.rand(B, N, device=dev, generator=g) ** 4, torch.rand(B, N, device=dev, generator=g) ** 4
    pos = torch.linspace(0, 1, N, device=dev)
    pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2)
    for flags, p in ((8, 1.0), (15, 2.0)):
        rows = nat.forward_rows(x, y, pos, pos2, p, flags, plan)
        want_mean, want_sum = nat.reduce_mean(rows, want_sum=True)
        for rep in range(4):
            mean, rows2, total = nat.loss_fused(x, y, pos, pos2, p, flags, plan, want_sum=True, fused_mean=True)
            assert torch.equal(mean, want_mean) and torch.equal(total, want_sum) and torch.equal(rows2, rows)
        two_kernels = nat.loss_fused(x, y, pos, pos2, p, flags, plan, fused_mean=False)[0]
        assert torch.equal(two_kernels, want_mean)
        h = float(rows.median())
        want_h = nat.reduce_mean(rows, hinge=h)
        assert torch.equal(nat.loss_fused(x, y, pos, pos2, p, flags, plan, hinge=h, fused_mean=True)[0], want_h)
        m1, r1, g1 = nat.loss_and_grad(x, y, pos, pos2, p, flags, plan, fused_mean=True)
        m2, r2, g2 = nat.loss_and_grad(x, y, pos, pos2, p, flags, plan, fused_mean=False)
        assert torch.equal(m1, m2) and torch.equal(r1, r2) and torch.equal(g1, g2) and torch.equal(m1, nat.reduce_mean(r1))
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for s in (s1, s2, s1, s2, s1, s2):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                outs.append(nat.loss_fused(x, y, pos, pos2, p, flags, plan, fused_mean=True)[0])
        torch.cuda.synchronize()
        assert all(torch.equal(o, want_mean) for o in outs)
    pool = nat._counter_pools[dev.index if dev.index is not None else torch.cuda.current_device()][0]
    assert int(pool.abs().sum()) == 0   # every launch put its counters back to zero


@pytest.mark.gpu
def test_two_host_threads_share_the_library():
    """the entry points are safe to call from several Python threads (ctypes releases the GIL; autograd runs
    backward on its own thread): launch state (occupancy-sized grids, LDS opt-in) is per device and behind a mutex.  Two
    threads hammer different kernels, first use included, on their own streams; every result equals the single-threaded one."""
    import threading
    nat = native()
    dev = device()
    shapes = [(300, 257), (70, 2048), (33, 1025), (90, 300), (64, 512), (20, 4096)]
    g = torch.Generator(device=dev).manual_seed(1)
    data = []
    for B, N in shapes:
        x, y = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
        pos = torch.linspace(0, 1, N, device=dev)
        data.append((x, y, pos, pos.clone()))
    torch.cuda.synchronize()
    results = [[None] * len(shapes) for _ in range(2)]
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(dev)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                order = range(len(shapes)) if k == 0 else reversed(range(len(shapes)))
                for i in order:
                    x, y, pos, pos2 = data[i]
                    for _ in range(20):
                        rows = nat.forward_rows(x, y, pos, pos2, 2.0, 15)
                        gx, gy = nat.backward_rows(x, y, pos, pos2, 2.0, 15, torch.ones(1, device=dev), grad_scale=1.0)
                    results[k][i] = (rows.clone(), gy.clone())
            s.synchronize()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, (x, y, pos, pos2) in enumerate(data):
        rows = nat.forward_rows(x, y, pos, pos2, 2.0, 15)
        _, gy = nat.backward_rows(x, y, pos, pos2, 2.0, 15, torch.ones(1, device=dev), grad_scale=1.0)
        for k in range(2):
            assert torch.equal(results[k][i][0], rows) and torch.equal(results[k][i][1], gy)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "peaky"])
@pytest.mark.parametrize("mode", ["p1", "cutoff", "nocut"])
def test_config4_full_size_dense_scalars(kind, mode, manifest):
    """BASELINE config 4 at its real size (B=8192, N=512), dense form: the reference's scalars for seed 1234
    (tests/golden/manifest.json, written by oracle/make_golden.py from /root/reference/losses.py)."""
    from sot_amd.bench_inputs import spectrum_pairs
    want = manifest["_config4_dense_b8192n512_seed1234"][f"{kind}_{mode}"]
    ctor = {"p1": dict(p=1), "cutoff": dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True),
            "nocut": dict(p=2, square_dist=True)}[mode]
    x, y = spectrum_pairs(kind, 8192, 512, 512, 1234)
    mod = module_for(ctor)
    pos = torch.linspace(0, 1, 512, device=device())
    with torch.no_grad():
        got = float(mod(x.to(device()), y.to(device()), x_pos=pos, y_pos=pos.clone()))
    assert abs(got - want) <= 1e-5 * abs(want), (got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["p1", "cutoff"])
def test_config4_full_size_csr_equals_masked_dense(mode):
    """BASELINE config 4 at its real size, ragged form: per-row amplitude cutoff -> variable supports (mean 187 of 512);
    the CSR kernel on the kept points and the dense kernel on the zero-masked rows give the same row losses (zero-weight
    points are inert), and a strided sample of rows agrees with the C oracle run on the truly ragged rows."""
    from oracle import sot_oracle as so
    from sot_amd.bench_inputs import ragged_supports
    from sot_amd.losses import wasserstein_1d_csr
    nat = native()
    dev = device()
    rs = ragged_supports(8192, 512, 1234)
    kw = {"p1": dict(p=1), "cutoff": dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)}[mode]
    flags = (1 if kw.get("square_dist") else 0) | (2 if kw.get("dont_normalize") else 0) | (4 if kw.get("limit_quantile_range") else 0) | 8
    xm, ym = (t.to(dev) for t in rs["dense"])
    (xw, xp, xo), (yw, yp, yo) = rs["csr"]
    pos = rs["pos"].to(dev)
    dense = nat.forward_rows(xm, ym, pos, pos.clone(), float(kw["p"]), flags)
    csr = wasserstein_1d_csr(xw.to(dev), xp.to(dev), xo.to(dev), yw.to(dev), yp.to(dev), yo.to(dev), rs["max_n"], rs["max_m"], **kw)
    torch.cuda.synchronize()
    d, c = dense.cpu().numpy().astype(np.float64), csr.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(c))
    # cutoff mode: removing the zero-weight points changes the ATen summation order of the row mass, hence (knife edge,
    # SURVEY B.1) single rows; the batch means agree to 1e-5 and almost every row to 1e-5
    rel = np.abs(d - c) / np.maximum(np.abs(d), 1e-30)
    flips = float(np.mean(rel > 1e-5))
    print(f"config 4 {mode}: batch means {d.mean():.9g} (masked dense) {c.mean():.9g} (CSR), rows differing by > 1e-5: {flips:.4f}, "
          f"median rel {np.median(rel):.2e}")
    if mode == "cutoff":
        # Two effects separate the forms in the paper's mode (round 3, found with the dyadic fixtures below):
        #  (1) by the reference's OWN semantics they are different problems under dont_normalize: the levels between V_last and
        #      U_last take ys[m - 1] (the clamp of losses.py:220) -- the grid's last point in the masked-dense rows, the last KEPT
        #      point in the ragged rows.  That is the bulk of the rows that differ, and it is not a rounding matter;
        #  (2) the knife edge: zeros removed -> another cascade order of torch.sum -> S differs by an ulp in part of
        #      the rows, and a level at U_last == 1 +- ulp carries percents of such a row's loss.
        # Keeping the grid's last point in the CSR rows (weight 0 where it fell below the threshold) removes (1); what is left is (2).
        assert np.median(rel) <= 1e-6 and flips <= 0.6 and abs(d.mean() - c.mean()) <= 2e-2 * abs(d.mean())
        kx, ky = xm.cpu() > 0, ym.cpu() > 0
        kx[:, -1] = True
        ky[:, -1] = True
        (aw, ap, ao), (bw, bp, bo) = _to_csr(xm.cpu(), rs["pos"], kx), _to_csr(ym.cpu(), rs["pos"], ky)
        anch = wasserstein_1d_csr(aw.to(dev), ap.to(dev), ao.to(dev), bw.to(dev), bp.to(dev), bo.to(dev), int(kx.sum(1).max()),
                                  int(ky.sum(1).max()), **kw).cpu().numpy().astype(np.float64)
        rel_a = np.abs(d - anch) / np.maximum(np.abs(d), 1e-30)
        flips_a = float(np.mean(rel_a > 1e-5))
        print(f"config 4 cutoff, CSR with the grid's last point kept: rows differing from masked dense by > 1e-5: {flips_a:.4f}, "
              f"batch means {d.mean():.9g} / {anch.mean():.9g}")
        assert flips_a <= 0.15 and abs(d.mean() - anch.mean()) <= 2e-3 * abs(d.mean())   # the knife edge alone (observed: see DESIGN section 6)
    else:
        assert abs(d.mean() - c.mean()) <= 1e-5 * abs(d.mean()) and flips == 0.0
    for r in range(0, 8192, 512):   # oracle on the ragged rows themselves
        a, b = int(xo[r]), int(xo[r + 1])
        e, f = int(yo[r]), int(yo[r + 1])
        want = so.forward(xw[a:b].numpy()[None], yw[e:f].numpy()[None], xp[a:b].numpy(), yp[e:f].numpy(), p=float(kw["p"]), flags=flags)[0]
        assert abs(c[r] - want) <= 1e-5 * abs(want), (r, c[r], want)


@pytest.mark.gpu
def test_config5_full_size_training_step_matches_reference():
    """BASELINE config 5 at its real size: 256 + 256 harmonic clips -> STFT (n_fft 2048, hop 256, flattop) -> 4096 rows x
    1025 bins -> SOT paper mode -> gradient into the estimate's audio, through spectra.training_step_slice (HIP STFT pair,
    loss-and-gradient kernel, HIP STFT backward).  Expected values: the reference's own TorchSTFT + Wasserstein1D + autograd
    on the same clips (oracle/make_golden_config5.py -> tests/golden/config5_256.npz); the clips are regenerated bit for bit
    (sha256 checked)."""
    from oracle.inputs import exact_harmonic_clips, sha256_of
    from sot_amd import spectra
    fx = np.load(os.path.join(GOLDEN, "config5_256.npz"))
    target, estimate = exact_harmonic_clips(int(fx["clips"]), int(fx["seed"]))
    assert sha256_of(target, estimate) == bytes(fx["inputs_sha256"]).hex(), "the seeded clips are not the fixture's"
    dev = device()
    mod = module_for(dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True))
    est = estimate.to(dev).requires_grad_(True)
    loss = spectra.training_step_slice(mod, target.to(dev), est)
    loss.backward()
    torch.cuda.synchronize()
    want = float(fx["loss"])
    assert abs(float(loss.detach()) - want) <= 2e-5 * abs(want), (float(loss.detach()), want)   # knife edge of the cutoff, SURVEY B.1
    stride = int(fx["stride"])
    got = est.grad[:, ::stride].cpu().numpy().astype(np.float64)
    ref = fx["grad_audio_sample"].astype(np.float64)
    # Which clips are clean?  The spectra of the two STFT implementations differ in the last bits (3e-7 of the peak); in the
    # paper's cutoff mode such a difference moves single rows by percents (a level at U_last = 1 +- ulp is kept or dropped,
    # SURVEY B.1) and with them the gradient of their clip.  Row losses tell the clips apart: a clip is clean when its 16
    # rows all agree with the reference's to 1e-6.
    with torch.no_grad():
        rows = mod.row_losses(spectra.stft_magnitude(target.to(dev)), spectra.stft_magnitude(estimate.to(dev)),
                              x_pos=spectra.unit_frequencies(2048, 16000.0, dev), y_pos=spectra.unit_frequencies(2048, 16000.0, dev).clone())
    rows = rows.cpu().numpy().astype(np.float64).reshape(256, 16)
    ref_rows = fx["row_loss"].astype(np.float64).reshape(256, 16)
    row_rel = np.abs(rows - ref_rows) / np.maximum(ref_rows, 1e-30)
    clean = (row_rel <= 1e-6).all(axis=1)
    clip_peak = np.abs(ref).max(axis=1, keepdims=True)
    err = np.abs(got - ref) / clip_peak
    cos = float((got * ref).sum() / np.sqrt((got * got).sum() * (ref * ref).sum()))
    print(f"config 5 @256 clips: loss rel {abs(float(loss.detach()) - want) / want:.2e}; rows off by > 1e-6: {np.mean(row_rel > 1e-6):.4f}; "
          f"clean clips {int(clean.sum())}/256: max grad err / clip peak {err[clean].max():.2e} (median {np.median(err[clean]):.2e}); "
          f"other clips: max {err[~clean].max() if (~clean).any() else 0:.2e}; overall cosine {cos:.6f}")
    assert clean.sum() >= 119                                 # observed 133-134 of 256 (rounds 2-3), 139 (round 4); minus 10 %
    # clean clips: the gradient agrees with the reference's autograd (observed: median 1.1e-6 of the clip's peak, 99th percentile 5e-5; the
    # maximum is a lottery of ties -- a tie between two levels may route a run's gradient to another member, a different valid
    # subgradient, DESIGN section 2 -- whose outcome depends on the last bits of the spectra: 3.9e-4 with the slot STFT kernels of
    # rounds 2-3 (134 clean clips), 1.35e-3 with the one-wave-per-frame forward kernel of round 4 (139 clean clips; the backward kernel
    # does not change it: round 4 GPU call 13, docs/HISTORY.md §13))
    assert np.median(err[clean]) <= 5e-6 and np.percentile(err[clean], 99) <= 1e-4
    # THE KERNELS' PARITY STATEMENT, over ALL 256 clips (round-5 review; rounds 4-5 did this for the clean clips' outliers only and held the other
    # 117 clips to a cosine): the whole batch is re-evaluated with the REFERENCE'S arithmetic on the HIP kernels' OWN spectra -- C oracle backward
    # (stable tie order) on the 4096 rows, then torch.stft's autograd on the CPU -- and the HIP gradient must agree with that to 1e-4 of each
    # clip's peak, clean or not: the cutoff's lottery (which rows a last-bit difference between two float32 STFTs flips) is then out of the
    # kernels' statement altogether, because both sides see the same spectra.
    from oracle import sot_oracle as so
    with torch.no_grad():
        hip_t = spectra.stft_magnitude(target.to(dev)).cpu().numpy().reshape(4096, 1025)
        hip_e = spectra.stft_magnitude(estimate.to(dev)).cpu().numpy().reshape(4096, 1025)
    pos_np = spectra.unit_frequencies(2048, 16000.0, "cpu").numpy()
    oflags = so.make_flags(True, True, True, True)
    _, gspec = so.backward(hip_t, hip_e, pos_np, pos_np, np.full(4096, 1.0 / 4096, np.float32), p=2.0, flags=oflags)
    e_cpu = estimate.clone().requires_grad_(True)
    (spectra.stft_magnitude_torch(e_cpu) * torch.as_tensor(gspec).reshape(256, 16, 1025)).sum().backward()
    via_ref_full = e_cpu.grad.numpy().astype(np.float64)
    got_full = est.grad.cpu().numpy().astype(np.float64)
    peak_full = np.abs(via_ref_full).max(axis=1, keepdims=True)
    own = np.abs(got_full - via_ref_full).max(axis=1) / peak_full[:, 0]
    print(f"config 5, all 256 clips, HIP vs the reference's arithmetic on the HIP spectra: max {own.max():.2e}, median {np.median(own):.2e} of the clip's peak "
          f"(clean clips max {own[clean].max():.2e}, the other {int((~clean).sum())} max {own[~clean].max() if (~clean).any() else 0:.2e})")
    assert own.max() <= 1e-4, (int(own.argmax()), float(own.max()))
    # Second: against the reference FIXTURE (its own spectra).  What remains between the two is reference arithmetic on HIP spectra vs reference
    # arithmetic on torch.stft spectra (two float32 STFTs, 3e-7 apart): the conditioning of the reference's gradient itself -- a 1-ulp perturbation
    # of its own spectra that leaves every row loss within 1e-6 moves its audio gradient by up to 4.5e-3 of a clip's peak (median 3.9e-6; measured
    # with the oracle over 64 clips x 3 seeds, round 5).  On the clean clips every outlier beyond 1e-4 must BE that conditioning.
    via_ref = via_ref_full[:, ::stride]
    cond = np.abs(via_ref - ref).max(axis=1) / clip_peak[:, 0]
    outliers = [int(c) for c in np.nonzero(clean)[0] if err[c].max() > 1e-4]
    assert len(outliers) <= 24, len(outliers)
    for c in outliers:
        assert abs(float(err[c].max()) - cond[c]) <= 2e-4, (c, float(err[c].max()), float(cond[c]))
    print(f"config 5 vs the fixture: {len(outliers)} clean clips beyond 1e-4, each explained by the reference's conditioning (<= {max([cond[c] for c in outliers], default=0.0):.2e}); "
          f"overall cosine {cos:.6f} (reported, not asserted: flipped rows move single clips)")
    assert err[clean].max() <= 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("N,B", [(129, 700), (257, 530), (512, 1030), (513, 300), (1024, 333), (1025, 261), (2048, 520), (2049, 77), (4096, 70),
                                 (8192, 35)])
@pytest.mark.parametrize("flags", [0, 1, 2, 1 | 2])
def test_p1_same_grid_merge_free_kernel(N, B, flags):
    """p = 1 with both measures on one grid (SOT_FLAG_SAME_GRID, set by the binding from the position plan): the forward
    evaluates sum_i |U_i - V_i| (pos_{i+1} - pos_i) instead of merging the CDFs (losses.py:295-313).  Same value: against the
    merge kernel (SOT_FLAG_NO_AREA) to 2e-6 per row, against the C oracle to 1e-5, peaky and uniform rows, square_dist and
    dont_normalize (total masses differ: the clamp of losses.py:220 matters), odd row lengths, a non-uniform grid."""
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    nat = native()
    dev = device()
    for kind in ("peaky", "uniform"):
        x, y = gen_inputs(kind, B, N, N, N + B)
        x[0] = 0.0                                   # zero row (mass guard)
        y[1] = x[1]                                  # identical distributions -> 0
        pos = torch.sort(torch.rand(N, generator=torch.Generator().manual_seed(N)) ** 2).values   # non-uniform grid
        xd, yd, pd = x.to(dev), y.to(dev), pos.to(dev)
        pd2 = pd.clone()
        plan = nat.PositionPlan(pd, pd2)
        assert plan.same_grid()
        f = flags | nat.FLAG_REQUIRE_SORT
        area = nat.forward_rows(xd, yd, pd, pd2, 1.0, f, plan).cpu().numpy().astype(np.float64)
        merge = nat.forward_rows(xd, yd, pd, pd2, 1.0, f | nat.FLAG_NO_AREA, plan).cpu().numpy().astype(np.float64)
        want = so.forward(x.numpy(), y.numpy(), pos.numpy(), pos.numpy(), p=1.0, flags=f).astype(np.float64)
        scale = np.maximum(np.abs(want), 1e-12)
        assert np.all(np.abs(area - merge) <= 2e-6 * np.maximum(np.abs(merge), 1e-12) + 1e-12), float(np.max(np.abs(area - merge) / scale))
        assert np.all(np.abs(area - want) <= 1e-5 * scale + 1e-12), float(np.max(np.abs(area - want) / scale))
        assert area[1] == 0.0


@pytest.mark.gpu
def test_p1_same_grid_dispatch_conditions():
    """The merge-free kernel is only taken when it applies: p == 1, no quantile cutoff, one grid for both measures.  Unsorted
    shared positions (same values both sides) still qualify after the plan's sort; different grids, p = 2 and the cutoff
    keep the merge kernel (results then equal the NO_AREA ones bit for bit)."""
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    from sot_amd.losses import Wasserstein1D
    nat = native()
    dev = device()
    B, N = 64, 2048
    x, y = gen_inputs("peaky", B, N, N, 77)
    xd, yd = x.to(dev), y.to(dev)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
    pos = torch.linspace(0, 1, N)[perm]                      # unsorted, same on both sides
    pd, pd2 = pos.to(dev), pos.to(dev).clone()
    plan = nat.PositionPlan(pd, pd2)
    assert plan.same_grid()
    got = nat.forward_rows(xd, yd, pd, pd2, 1.0, 8, plan).cpu().numpy()
    want = so.forward(x.numpy(), y.numpy(), pos.numpy(), pos.numpy(), p=1.0, flags=8)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-12)
    other = torch.sort(torch.rand(N, generator=torch.Generator().manual_seed(2))).values.to(dev)
    plan2 = nat.PositionPlan(torch.linspace(0, 1, N, device=dev), other)
    assert not plan2.same_grid()
    lin = torch.linspace(0, 1, N, device=dev)
    plan3 = nat.PositionPlan(lin, lin.clone())
    for p, f in ((2.0, 8), (1.0, 8 | 4), (2.0, 15)):         # p = 2 / cutoff: the flag changes nothing
        a = nat.forward_rows(xd, yd, lin, lin, p, f, plan3)
        b = nat.forward_rows(xd, yd, lin, lin, p, f | nat.FLAG_NO_AREA, plan3)
        assert torch.equal(a, b)
    # the module (metrics.py:148 form: fixed_x grid, p = 1) goes through the plan and therefore through the merge-free kernel
    mod = Wasserstein1D(p=1, fixed_x=N).to(dev)
    with torch.no_grad():
        m = float(mod(xd, yd))
    w = float(np.mean(so.forward(x.numpy(), y.numpy(), np.linspace(0, 1, N, dtype=np.float32), np.linspace(0, 1, N, dtype=np.float32), p=1.0,
                                 flags=8).astype(np.float64)))
    assert abs(m - w) <= 1e-5 * abs(w)


@pytest.mark.parametrize("mode", ["p1", "nocut", "cutoff"])
def test_csr_backward_matches_oracle_on_the_ragged_rows(mode):
    """Gradients of the CSR form w.r.t. the kept weights (missing #5): against the oracle's backward evaluated on
    the truly ragged rows, and -- where no knife edge interferes -- against the masked-dense module's gradient at the kept points."""
    from oracle import sot_oracle as so
    from oracle.inputs import gen_inputs
    from oracle.make_golden import MODES
    from sot_amd.losses import wasserstein_1d_csr
    native()
    dev = device()
    B, N = 96, 257
    x, y = gen_inputs("peaky", B, N, N, 77)
    g = torch.Generator().manual_seed(77)
    tau = 10 ** (-3 + 2.7 * torch.rand(B, 1, generator=g))
    kx, ky = x >= tau * x.amax(1, keepdim=True), y >= tau * y.amax(1, keepdim=True)
    pos = torch.linspace(0, 1, N)
    ctor = MODES[mode]
    p, flags = ctor_to_flags(ctor)
    xw, xp, xo = _to_csr(x, pos, kx)
    yw, yp, yo = _to_csr(y, pos, ky)
    max_n, max_m = int(kx.sum(1).max()), int(ky.sum(1).max())
    kw = dict(p=ctor.get("p", 1), square_dist=ctor.get("square_dist", False), dont_normalize=ctor.get("dont_normalize", False),
              limit_quantile_range=ctor.get("limit_quantile_range", False))
    xwd, ywd = xw.to(dev).requires_grad_(True), yw.to(dev).requires_grad_(True)
    upstream = torch.rand(B, generator=g)
    rows = wasserstein_1d_csr(xwd, xp.to(dev), xo.to(dev), ywd, yp.to(dev), yo.to(dev), max_n, max_m, **kw)
    (rows * upstream.to(dev)).sum().backward()
    gx, gy = xwd.grad.cpu().numpy(), ywd.grad.cpu().numpy()
    assert gx.shape == (int(xo[-1]),) and gy.shape == (int(yo[-1]),) and np.isfinite(gx).all() and np.isfinite(gy).all()
    checked = 0
    for r in range(0, B, 6):
        xr, yr = x[r][kx[r]][None].numpy(), y[r][ky[r]][None].numpy()
        wx, wy = so.backward(xr, yr, pos[kx[r]].numpy(), pos[ky[r]].numpy(), upstream[r:r + 1].numpy(), p=p, flags=flags)
        a, b = gx[int(xo[r]):int(xo[r + 1])], gy[int(yo[r]):int(yo[r + 1])]
        scale = max(np.abs(wx).max(), np.abs(wy).max(), 1e-30)
        if ctor.get("limit_quantile_range", False):
            # the padded rows' mass is summed over other operands than the ragged rows' (torch.sum order): rows on the cutoff's
            # knife edge may differ; the bulk must agree
            checked += int(np.abs(a - wx[0]).max() <= 2e-4 * scale and np.abs(b - wy[0]).max() <= 2e-4 * scale)
        else:
            assert np.abs(a - wx[0]).max() <= 2e-5 * scale and np.abs(b - wy[0]).max() <= 2e-5 * scale, r
            checked += 1
    assert checked >= 10
    # only one side needs a gradient; none at all -> the plain forward
    yw2 = yw.to(dev).requires_grad_(True)
    wasserstein_1d_csr(xw.to(dev), xp.to(dev), xo.to(dev), yw2, yp.to(dev), yo.to(dev), max_n, max_m, **kw).sum().backward()
    assert yw2.grad.shape == yw.shape
    with torch.no_grad():
        again = wasserstein_1d_csr(xw.to(dev), xp.to(dev), xo.to(dev), yw2, yp.to(dev), yo.to(dev), max_n, max_m, **kw)
    assert torch.equal(again, rows.detach())


# ---------------------------------------------------------------------------------------------------------------------------
# Dyadic full-size fixtures (SURVEY Appendix B.1 iii; oracle/make_golden_dyadic.py -> tests/golden/dyadic_full_size.npz): weights
# k/32 sum exactly in float32 in any order, so the row mass and the cutoff's knife edge are the same for the reference, for the
# dense kernels and for the CSR form -- the comparisons that are "bulk only" on random inputs are row for row here.
# ---------------------------------------------------------------------------------------------------------------------------
def _dyadic_fixture():
    return np.load(os.path.join(GOLDEN, "dyadic_full_size.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cutoff", "p1"])
def test_config4_dyadic_dense_and_csr_match_the_reference_row_for_row(mode):
    """BASELINE config 4 at full size (8192 x 512, ragged supports) on dyadic weights, against the reference's 8192 row losses
    (computed on the zero-masked dense rows): EVERY row within 1e-5, cutoff mode included, for
      * the masked-dense rows through the dense kernel,
      * the CSR kernel on the kept points plus the grid's last point (kept as a zero-weight point when it fell below the row's
        threshold): in dont_normalize mode the levels between V_last and U_last take ys[m - 1] (the clamp of losses.py:220), so
        the last point of the support is part of the problem -- the plain ragged form (last KEPT point) is a different problem
        there and is pinned to the oracle on the ragged rows themselves (every 8th row);
      * p = 1 (both CDFs end at the same mass): the plain ragged CSR form as well."""
    from oracle import sot_oracle as so
    from oracle.inputs import sha256_of
    from sot_amd.bench_inputs import dyadic_ragged_supports
    from sot_amd.losses import wasserstein_1d_csr
    nat = native()
    dev = device()
    fx = _dyadic_fixture()
    rs = dyadic_ragged_supports(8192, 512, int(fx["seed"]))
    xm, ym = rs["dense"]
    assert sha256_of(xm, ym) == bytes(fx["c4_inputs_sha256"]).hex(), "the seeded dyadic rows are not the fixture's"
    kw = {"p1": dict(p=1), "cutoff": dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)}[mode]
    flags = (1 if kw.get("square_dist") else 0) | (2 if kw.get("dont_normalize") else 0) | (4 if kw.get("limit_quantile_range") else 0) | 8
    pos = rs["pos"]
    (xw, xp, xo), (yw, yp, yo) = rs["csr"]
    want = fx[f"c4_{mode}_rows"].astype(np.float64)
    scale = np.maximum(np.abs(want), 1e-30)

    def csr_rows(xw_, xp_, xo_, yw_, yp_, yo_):
        mn, mm = int((xo_[1:] - xo_[:-1]).max()), int((yo_[1:] - yo_[:-1]).max())
        return wasserstein_1d_csr(xw_.to(dev), xp_.to(dev), xo_.to(dev), yw_.to(dev), yp_.to(dev), yo_.to(dev), mn, mm,
                                  **kw).cpu().numpy().astype(np.float64)

    dense = nat.forward_rows(xm.to(dev), ym.to(dev), pos.to(dev), pos.to(dev).clone(), float(kw["p"]), flags).cpu().numpy().astype(np.float64)
    kx, ky = xm > 0, ym > 0
    kx[:, -1] = True
    ky[:, -1] = True
    anchored = csr_rows(*_to_csr(xm, pos, kx), *_to_csr(ym, pos, ky))
    ragged = csr_rows(xw, xp, xo, yw, yp, yo)
    rel_d, rel_a = np.abs(dense - want) / scale, np.abs(anchored - want) / scale
    print(f"config 4 dyadic {mode}: dense max rel {rel_d.max():.2e}, CSR (last grid point kept) max rel {rel_a.max():.2e}, "
          f"plain ragged CSR vs dense: rows differing by > 1e-5: {np.mean(np.abs(ragged - dense) / scale > 1e-5):.4f}")
    assert rel_d.max() <= 1e-5, float(rel_d.max())          # 100 % of the rows
    assert rel_a.max() <= 1e-5, float(rel_a.max())
    assert abs(dense.mean() - float(fx[f"c4_{mode}_scalar"])) <= 2e-6 * abs(dense.mean())
    if mode == "p1":
        assert (np.abs(ragged - want) / scale).max() <= 1e-5
    for r in range(0, 8192, 8):   # the plain ragged form against the oracle on the ragged rows themselves
        a, b, e, f = int(xo[r]), int(xo[r + 1]), int(yo[r]), int(yo[r + 1])
        w = so.forward(xw[a:b].numpy()[None], yw[e:f].numpy()[None], xp[a:b].numpy(), yp[e:f].numpy(), p=float(kw["p"]), flags=flags)[0]
        assert abs(ragged[r] - w) <= 1e-5 * abs(w), (r, ragged[r], w)


@pytest.mark.gpu
def test_config5_sot_stage_dyadic_matches_the_reference_row_for_row():
    """The SOT stage of BASELINE config 5 at its real size (4096 rows x 1025 bins, rfftfreq positions, paper mode) on dyadic
    spectra: loss, ALL 4096 row losses (<= 1e-5 each: no knife-edge excuse when the row mass is order-independent) and the
    gradient w.r.t. y -- against the oracle's closed form on a sample of rows (<= 1e-5 of the row's largest entry), and against
    the reference's autograd sample wherever its unstable level sort happened to agree with the stable convention."""
    from oracle import sot_oracle as so
    from oracle.inputs import sha256_of
    from sot_amd import spectra
    from sot_amd.bench_inputs import dyadic_pairs
    fx = _dyadic_fixture()
    x, y = dyadic_pairs(4096, 1025, int(fx["seed"]) + 1)
    assert sha256_of(x, y) == bytes(fx["c5_inputs_sha256"]).hex()
    dev = device()
    ctor = dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
    mod = module_for(ctor)
    pos = spectra.unit_frequencies(2048, 16000.0, dev)
    yd = y.to(dev).requires_grad_(True)
    loss = mod(x.to(dev), yd, x_pos=pos, y_pos=pos.clone())
    loss.backward()
    want = float(fx["c5_scalar"])
    assert abs(float(loss.detach()) - want) <= 2e-6 * abs(want), (float(loss.detach()), want)
    with torch.no_grad():
        rows = mod.row_losses(x.to(dev), y.to(dev), x_pos=pos, y_pos=pos.clone()).cpu().numpy().astype(np.float64)
    ref_rows = fx["c5_rows"].astype(np.float64)
    rel = np.abs(rows - ref_rows) / np.maximum(ref_rows, 1e-30)
    assert rel.max() <= 1e-5, float(rel.max())               # 100 % of the rows
    got = yd.grad.cpu().numpy()
    p, flags = ctor_to_flags(ctor)
    pn = pos.cpu().numpy()
    for r in range(0, 4096, 64):
        _, gy = so.backward(x[r:r + 1].numpy(), y[r:r + 1].numpy(), pn, pn, np.full(1, 1.0 / 4096, np.float32), p=p, flags=flags)
        assert np.abs(got[r] - gy[0]).max() <= 1e-5 * np.abs(gy[0]).max(), r
    stride = int(fx["stride"])
    ref = fx["c5_grad_y_sample"].astype(np.float64)
    err = np.abs(got[:, ::stride] - ref) / fx["c5_grad_y_rowmax"].astype(np.float64)[:, None]
    print(f"config 5 SOT stage, dyadic: rows max rel {rel.max():.2e}; gradient vs the reference's autograd sample: median {np.median(err):.2e}, "
          f"entries within 1e-5 of the row max: {np.mean(err <= 1e-5):.4f}")
    assert np.median(err) <= 1e-6   # ties (zero-weight bins) may route a run's gradient elsewhere in the reference's unstable sort


@pytest.mark.gpu
def test_csr_backward_dyadic_cutoff_matches_the_oracle_on_every_checked_row():
    """test_csr_backward_matches_oracle_on_the_ragged_rows[cutoff] without the knife-edge allowance: dyadic weights make the padded
    rows' mass equal the ragged rows' mass exactly, so EVERY checked row must agree with the oracle's backward."""
    from oracle import sot_oracle as so
    from sot_amd.bench_inputs import dyadic_pairs
    from sot_amd.losses import wasserstein_1d_csr
    native()
    dev = device()
    B, N = 96, 257
    x, y = dyadic_pairs(B, N, 91)
    g = torch.Generator().manual_seed(91)
    tau = 10 ** (-3 + 2.7 * torch.rand(B, 1, generator=g))
    kx, ky = (x >= tau * x.amax(1, keepdim=True)) & (x > 0), (y >= tau * y.amax(1, keepdim=True)) & (y > 0)
    pos = torch.linspace(0, 1, N)
    ctor = dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
    p, flags = ctor_to_flags(ctor)
    xw, xp, xo = _to_csr(x, pos, kx)
    yw, yp, yo = _to_csr(y, pos, ky)
    max_n, max_m = int(kx.sum(1).max()), int(ky.sum(1).max())
    xwd, ywd = xw.to(dev).requires_grad_(True), yw.to(dev).requires_grad_(True)
    upstream = torch.rand(B, generator=g)
    rows = wasserstein_1d_csr(xwd, xp.to(dev), xo.to(dev), ywd, yp.to(dev), yo.to(dev), max_n, max_m, **ctor)
    (rows * upstream.to(dev)).sum().backward()
    gx, gy = xwd.grad.cpu().numpy(), ywd.grad.cpu().numpy()
    for r in range(B):
        xr, yr = x[r][kx[r]][None].numpy(), y[r][ky[r]][None].numpy()
        want_row = so.forward(xr, yr, pos[kx[r]].numpy(), pos[ky[r]].numpy(), p=p, flags=flags)[0]
        assert abs(float(rows[r]) - want_row) <= 1e-5 * abs(want_row), r
        wx, wy = so.backward(xr, yr, pos[kx[r]].numpy(), pos[ky[r]].numpy(), upstream[r:r + 1].numpy(), p=p, flags=flags)
        a, b = gx[int(xo[r]):int(xo[r + 1])], gy[int(yo[r]):int(yo[r + 1])]
        scale = max(np.abs(wx).max(), np.abs(wy).max(), 1e-30)
        assert np.abs(a - wx[0]).max() <= 2e-5 * scale and np.abs(b - wy[0]).max() <= 2e-5 * scale, r


# ---------------------------------------------------------------------------------------------------------------------------
# Run-time row lengths on the compile-time geometries (round 3; csrc/sot_forward_full.inc, NX = -1): every n == m <= 8192 on
# shared positions that has no kernel of its own runs the next capacity's geometry, its tail padded with zero-weight points at
# the row's last position.  Same results as the generic kernels (SOT_FLAG_NO_SPECIALIZE) and as the oracle.
# ---------------------------------------------------------------------------------------------------------------------------
RT_LENGTHS = [(130, 300), (200, 129), (256, 77), (300, 130), (500, 65), (511, 40), (640, 33), (1000, 70), (1023, 9), (1100, 21),
              (1536, 9), (1537, 9), (2000, 37), (2047, 5), (3000, 11), (3072, 4), (3073, 4), (4095, 3), (5000, 7), (8191, 2)]


@pytest.mark.parametrize("N,B", RT_LENGTHS)
@pytest.mark.parametrize("flags,p", [(0, 1.0), (1 | 2 | 4, 2.0), (1 | 4 | 8, 2.0), (2, 1.0), (1, 3.0)])
def test_runtime_length_forward_matches_generic_and_oracle(N, B, flags, p):
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    dev = device()
    x, y = gen_inputs("peaky", B, N, N, 700 + N)
    x, y = x.to(dev), y.to(dev)
    pos = torch.sort(torch.rand(N, generator=torch.Generator().manual_seed(N))).values if N % 2 else torch.linspace(0, 1, N)
    pos = pos.to(dev)
    pos2 = pos.clone() * 0.97 + 0.01          # two different grids: the merge kernel
    plan = nat.PositionPlan(pos, pos2) if flags & 8 else None
    spec = nat.forward_rows(x, y, pos, pos2, p, flags, plan)
    gen = nat.forward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, plan)
    torch.testing.assert_close(spec, gen, rtol=2e-6, atol=1e-12)
    k = B   # every row against the oracle
    want = so.forward(x[:k].cpu().numpy(), y[:k].cpu().numpy(), pos.cpu().numpy(), pos2.cpu().numpy(), p=p, flags=flags & 15)
    np.testing.assert_allclose(spec[:k].cpu().numpy(), want, rtol=RTOL)
    if p == 1.0 and not (flags & 4):          # one grid on both sides: the merge-free kernel with a run-time length
        plan1 = nat.PositionPlan(pos, pos.clone())
        area = nat.forward_rows(x, y, pos, pos, 1.0, flags | 8, plan1)
        merge = nat.forward_rows(x, y, pos, pos, 1.0, flags | 8 | nat.FLAG_NO_AREA, plan1)
        torch.testing.assert_close(area, merge, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("N,B", [(130, 60), (200, 33), (300, 41), (500, 19), (1000, 23), (1100, 9), (1536, 7), (1537, 5), (2000, 13), (3000, 5), (3072, 3),
                                 (3073, 3), (4095, 3)])
@pytest.mark.parametrize("flags,p", [(0, 1.0), (1 | 2 | 4, 2.0), (1 | 4 | 8, 2.0), (2, 1.0), (1, 3.0)])
def test_runtime_length_backward_and_training_form(N, B, flags, p):
    """Gradients (both, and y only) against the generic kernels and the oracle; the training form's row losses equal the
    forward kernel's bit for bit and its gradient equals the y-only backward's."""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    dev = device()
    x, y = gen_inputs("peaky", B, N, N, 900 + N)
    x, y = x.to(dev), y.to(dev)
    pos = torch.linspace(0, 1, N).to(dev)
    pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2) if flags & 8 else None
    g = torch.linspace(0.5, 1.5, B).to(dev)
    sx, sy = nat.backward_rows(x, y, pos, pos2, p, flags, g, plan=plan, grad_scale=0.5)
    gx, gy = nat.backward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_SPECIALIZE, g, plan=plan, grad_scale=0.5)
    k = B   # every row against the oracle
    wx, wy = so.backward(x[:k].cpu().numpy(), y[:k].cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(), (0.5 * g[:k]).cpu().numpy(),
                         p=p, flags=flags & 15)
    for got, ref, want in ((sx, gx, wx), (sy, gy, wy)):
        scale = ref.abs().amax(dim=1, keepdim=True) + 1e-30
        assert float(((got - ref).abs() / scale).max()) <= 2e-6
        wscale = np.abs(want).max(axis=1, keepdims=True) + 1e-30
        assert np.max(np.abs(got[:k].cpu().numpy() - want) / wscale) <= 1e-5
    only_y = nat.backward_rows(x, y, pos, pos2, p, flags, g, need_gx=False, plan=plan, grad_scale=0.5)
    assert only_y[0] is None and torch.equal(only_y[1], sy)
    rows = nat.forward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_AREA, plan)
    mean, rows2, gy2 = nat.loss_and_grad(x, y, pos, pos2, p, flags | nat.FLAG_NO_AREA, plan)
    assert torch.equal(rows, rows2)
    ones = nat.backward_rows(x, y, pos, pos2, p, flags, torch.ones(1, device=dev), need_gx=False, plan=plan, grad_scale=1.0 / B)[1]
    assert torch.equal(gy2, ones)


def test_runtime_length_unsorted_positions_strides_and_module():
    """Unsorted shared positions (the plan's permutation), strided rows, a 3-D batch through the module and autograd."""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    from sot_amd.losses import Wasserstein1D
    nat = native()
    dev = device()
    B, N = 50, 1000
    x, y = gen_inputs("peaky", B, N, N, 31)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(2))
    pos = torch.linspace(0, 1, N)[perm]
    big_x = torch.zeros(B, N + 24, device=dev)
    big_x[:, 3:N + 3] = x.to(dev)
    xv = big_x[:, 3:N + 3]                      # unaligned, strided rows
    yd = y.to(dev)
    pd, pd2 = pos.to(dev), pos.to(dev).clone()
    plan = nat.PositionPlan(pd, pd2)
    for p, flags in ((2.0, 15), (1.0, 8)):
        got = nat.forward_rows(nat.rows_view(xv), yd, pd, pd2, p, flags, plan).cpu().numpy()
        want = so.forward(x.numpy(), y.numpy(), pos.numpy(), pos.numpy(), p=p, flags=flags)
        np.testing.assert_allclose(got, want, rtol=RTOL)
        _, gy = nat.backward_rows(nat.rows_view(xv), yd, pd, pd2, p, flags, torch.ones(1, device=dev), need_gx=False, plan=plan)
        _, wy = so.backward(x[:4].numpy(), y[:4].numpy(), pos.numpy(), pos.numpy(), np.ones(4, np.float32), p=p, flags=flags)
        assert np.max(np.abs(gy[:4].cpu().numpy() - wy) / (np.abs(wy).max(axis=1, keepdims=True) + 1e-30)) <= 1e-5
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
    y3 = yd.reshape(5, 10, N).clone().requires_grad_(True)
    loss = mod(x.to(dev).reshape(5, 10, N), y3, x_pos=pd, y_pos=pd2)
    loss.backward()
    want = so.mean(so.forward(x.numpy(), y.numpy(), pos.numpy(), pos.numpy(), p=2.0, flags=15))
    assert abs(float(loss.detach()) - float(want)) <= 1e-5 * abs(float(want))
    assert torch.isfinite(y3.grad).all() and float(y3.grad.abs().max()) > 0


@pytest.mark.parametrize("N,B", [(1025, 6144), (1025, 9001), (1000, 8192), (777, 8200), (257, 40961), (129, 8195)])
@pytest.mark.parametrize("flags,p", [(1 | 2 | 4 | 8, 2.0), (8, 2.0), (0, 3.0)])
def test_large_batches_one_wave_per_row_forward(N, B, flags, p):
    """Batches of >= 6144 1025-bin rows (>= 8192 rows of any other length up to 1024) run the merge forward with ONE wave per row
    (17 / 16 elements per thread); large batches of 257- / 129-bin rows with TWO rows per wave (32 lanes x 9 / 5 elements).  Its row losses agree with the oracle like every other kernel's, and with the two-wave kernel's (the
    same rows as a small batch) to rounding -- its thread-local sums group the row differently, so not bit for bit."""
    from oracle.inputs import gen_inputs
    from oracle import sot_oracle as so
    nat = native()
    dev = device()
    x, y = gen_inputs("peaky", B, N, N, 31 + N)
    x, y = x.to(dev), y.to(dev)
    pos = torch.linspace(0, 1, N).to(dev)
    pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2) if flags & 8 else None
    big = nat.forward_rows(x, y, pos, pos2, p, flags | nat.FLAG_NO_AREA, plan)
    k = B   # every row against the oracle
    want = so.forward(x[:k].cpu().numpy(), y[:k].cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(), p=p, flags=flags & 15)
    np.testing.assert_allclose(big[:k].cpu().numpy(), want, rtol=RTOL)
    small = torch.cat([nat.forward_rows(x[i:i + 2000], y[i:i + 2000], pos, pos2, p, flags | nat.FLAG_NO_AREA, plan) for i in range(0, B, 2000)])
    torch.testing.assert_close(big, small, rtol=2e-6, atol=1e-12)
    last = nat.forward_rows(x[B - 3:], y[B - 3:], pos, pos2, p, flags | nat.FLAG_NO_AREA, plan)     # the end of the batch (partial workgroup)
    torch.testing.assert_close(big[B - 3:], last, rtol=2e-6, atol=1e-12)


def test_one_wave_rows_with_unsorted_positions_strides_and_the_in_kernel_mean():
    """The large-batch one-wave forward kernels (1025 bins, run-time lengths up to 1024) on a permuted grid, on rows that are views into
    wider buffers, through the module, and with the batch mean taken by the kernel's last workgroup."""
    from sot_amd.losses import Wasserstein1D
    nat = native()
    dev = device()
    for N, B in ((1025, 6200), (900, 8300)):
        g = torch.Generator().manual_seed(N)
        perm = torch.randperm(N, generator=g)
        pos_sorted = torch.linspace(0, 1, N)
        pos = pos_sorted[perm].contiguous()                     # unsorted shared grid: the plan carries the permutation
        wide_x, wide_y = torch.rand(B, N + 7, generator=g) ** 4, torch.rand(B, N + 7, generator=g) ** 4
        xs, ys = wide_x.to(dev)[:, 3:3 + N], wide_y.to(dev)[:, 5:5 + N]          # strided, 4-B aligned only
        mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
        pd_, pd2 = pos.to(dev), pos.to(dev).clone()
        rows_big = mod.row_losses(xs, ys, x_pos=pd_, y_pos=pd2)
        rows_small = torch.cat([mod.row_losses(xs[i:i + 1500], ys[i:i + 1500], x_pos=pd_, y_pos=pd2) for i in range(0, B, 1500)])
        torch.testing.assert_close(rows_big, rows_small, rtol=2e-6, atol=1e-12)
        # against the oracle on the same permuted grid (it sorts like the reference)
        from oracle import sot_oracle as so
        k = B   # every row against the oracle
        want = so.forward(xs[:k].cpu().numpy(), ys[:k].cpu().numpy(), pos.numpy(), pos.numpy(), p=2.0, flags=1 | 2 | 4 | 8)
        np.testing.assert_allclose(rows_big[:k].cpu().numpy(), want, rtol=RTOL)
        # the same rows on the sorted grid, columns permuted accordingly: the same problem up to the summation order of the row mass
        # (no cutoff here: in the paper's mode a last-bit change of the mass moves rows across the cutoff's knife edge)
        plain = Wasserstein1D(p=2, square_dist=True).to(dev)
        inv = torch.argsort(perm)
        xs2, ys2 = xs[:, inv.to(dev)].contiguous(), ys[:, inv.to(dev)].contiguous()
        ps = pos_sorted.to(dev)
        torch.testing.assert_close(plain.row_losses(xs, ys, x_pos=pd_, y_pos=pd2), plain.row_losses(xs2, ys2, x_pos=ps, y_pos=ps.clone()),
                                   rtol=5e-5, atol=1e-12)   # (row masses of rand^8 weights summed in another order)
        plan = nat.PositionPlan(pd_, pd2)
        flags = 1 | 2 | 4 | 8
        two = nat.loss_fused(xs, ys, pd_, pd2, 2.0, flags, plan)
        one = nat.loss_fused(xs, ys, pd_, pd2, 2.0, flags, plan, fused_mean=True)
        assert torch.equal(two[0] if isinstance(two, tuple) else two, one[0] if isinstance(one, tuple) else one)


def test_unit_position_plan_equals_torch_division_and_sort():
    """sot_prepare_unit_positions (ABI 12): the plan of xpos / xpos.max() and ypos / ypos.max() out of one launch -- the floats of torch's
    max + true division (bit for bit), then the same sorted positions / permutations / identity flags as the plan of those quotients;
    sorted and unsorted grids, n != m, the paper's bin frequencies, a NaN (torch.max's rule: everything NaN), through the C++ host path
    and through ctypes."""
    nat = native()
    dev = device()
    g = torch.Generator().manual_seed(5)
    grids = [(torch.fft.rfftfreq(2048, d=1.0 / 16000.0), torch.fft.rfftfreq(2048, d=1.0 / 16000.0)),
             (torch.rand(1025, generator=g) * 8000.0, torch.rand(513, generator=g) * 3.0 + 0.1),
             (torch.linspace(0.5, 977.0, 300), torch.rand(300, generator=g) + 1e-3),
             (torch.tensor([3.0, 1.0, 2.0]), torch.tensor([7.0]))]
    import sot_amd._native as native_mod
    for use_glue in (True, False):
        saved = native_mod.glue
        if not use_glue:
            native_mod.glue = lambda: None
        try:
            for xf, yf in grids:
                xf, yf = xf.float().to(dev), yf.float().to(dev)
                unit = nat.PositionPlan(xf, yf, unit=True)
                want = nat.PositionPlan(xf / xf.max(), yf / yf.max())
                torch.cuda.synchronize()
                assert torch.equal(unit.xpos_sorted, want.xpos_sorted) and torch.equal(unit.ypos_sorted, want.ypos_sorted)
                assert torch.equal(unit.xperm, want.xperm) and torch.equal(unit.yperm, want.yperm) and torch.equal(unit.ident, want.ident)
                assert torch.equal(unit.xpos_sorted, torch.sort(xf / xf.max(), stable=True).values)
            bad = torch.tensor([1.0, float("nan"), 4.0, 2.0], device=dev)
            plan = nat.PositionPlan(bad, bad.clone(), unit=True)
            assert bool(torch.isnan(plan.xpos_sorted).all()) and bool(torch.isnan(plan.ypos_sorted).all())   # x / NaN
        finally:
            native_mod.glue = saved
