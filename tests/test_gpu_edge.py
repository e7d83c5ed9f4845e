"""-m gpu: edge cases of the boundary (sizes, module plumbing, error behaviour) through the public Python API."""
import numpy as np
import pytest
import torch

from gpu_util import device, native

pytestmark = pytest.mark.gpu


def _oracle_rows(x, y, pos, **kw):
    from oracle import sot_oracle as so
    flags = so.make_flags(kw.get("square_dist", False), kw.get("dont_normalize", False), kw.get("limit_quantile_range", False), True)
    return so.forward(x.cpu().numpy(), y.cpu().numpy(), pos.cpu().numpy(), pos.cpu().numpy(), p=float(kw.get("p", 1)), flags=flags)


@pytest.mark.parametrize("B", [1, 2, 3, 5, 255, 257, 1023, 1025, 4097])
@pytest.mark.parametrize("N", [1, 3, 8, 31, 33, 64, 65, 513, 1537, 2049])
def test_row_and_length_boundaries(B, N):
    """row counts around the workgroup/row-group multiples and lengths around every geometry switch (64x8, 128x12, 256x8, 1024x8)"""
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    if B * N > 3_000_000:
        pytest.skip("kept small: the oracle is single-threaded")
    g = torch.Generator().manual_seed(B * 7919 + N)
    x, y = torch.rand(B, N, generator=g) ** 3, torch.rand(B, N, generator=g) ** 3
    pos = torch.linspace(0, 1, N)
    kw = dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
    mod = Wasserstein1D(**kw).to(dev)
    rows = mod(x.to(dev)[:, None, :], y.to(dev)[:, None, :], x_pos=pos.to(dev), y_pos=pos.to(dev), dims=[1]).cpu().numpy()
    want = _oracle_rows(x, y, pos, **kw)
    np.testing.assert_allclose(rows, want, rtol=1e-5, atol=1e-12)


def test_many_short_rows():
    """1M rows of 257 bins: the persistent grid strides many times; checked on a sample and through properties"""
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    B, N = 1 << 20, 257
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand(B, N, device=dev, generator=g)
    y = torch.rand(B, N, device=dev, generator=g)
    pos = torch.linspace(0, 1, N, device=dev)
    mod = Wasserstein1D(p=1).to(dev)
    rows = mod(x[:, None, :], y[:, None, :], x_pos=pos, y_pos=pos, dims=[1])
    assert rows.shape == (B,) and torch.isfinite(rows).all() and float(rows.min()) >= 0
    idx = torch.arange(0, B, B // 64, device=dev)
    want = _oracle_rows(x[idx], y[idx], pos, p=1)
    np.testing.assert_allclose(rows[idx].cpu().numpy(), want, rtol=1e-5)
    scalar = mod(x, y, x_pos=pos, y_pos=pos)
    torch.testing.assert_close(scalar, rows.double().mean().float(), rtol=1e-6, atol=0)


def test_module_plumbing():
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    m = Wasserstein1D(p=2, fixed_x=64)
    assert m.fixed_x.device.type == "cpu"
    m = m.to(dev)
    assert m.fixed_x.is_cuda                                   # the buffer follows .to(device) (metrics.py:148)
    sd = m.state_dict()
    m2 = Wasserstein1D(p=2, fixed_x=64).to(dev)
    m2.load_state_dict(sd)
    x, y = torch.rand(4, 64, device=dev), torch.rand(4, 64, device=dev)
    assert torch.equal(m(x, y), m2(x, y))
    # empty batch: mean of zero rows is NaN in the reference too (torch.mean of an empty tensor)
    e = m(x[:0][:, None, :], y[:0][:, None, :], dims=[1])
    assert e.shape == (0,)
    # positions that require a gradient get one (test_position_gradients_match_the_reference_autograd)
    xp = torch.linspace(0, 1, 64, device=dev, requires_grad=True)
    yy = y.clone().requires_grad_(True)
    Wasserstein1D(p=1).to(dev)(x, yy, x_pos=xp, y_pos=xp.detach().clone()).backward()
    assert xp.grad is not None and xp.grad.shape == (64,) and yy.grad is not None
    # half precision is refused rather than silently upcast
    with pytest.raises(TypeError):
        m(x.half(), y.half())
    # mismatched feature sizes
    with pytest.raises(RuntimeError):
        Wasserstein1D(p=1).to(dev)(x, y, x_pos=xp.detach()[:10], y_pos=xp.detach())


def test_gradients_only_where_requested():
    """trainer.py differentiates only w.r.t. the estimate (y): grad_x must not be computed or allocated"""
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    x = torch.rand(16, 1025, device=dev)
    y = torch.rand(16, 1025, device=dev, requires_grad=True)
    pos = torch.linspace(0, 1, 1025, device=dev)
    mod = Wasserstein1D(p=2, square_dist=True).to(dev)
    (mod(x, y, x_pos=pos, y_pos=pos) * 3.0).backward()       # MixOfLosses-style weight on the loss
    assert x.grad is None and y.grad is not None and torch.isfinite(y.grad).all()
    g3 = y.grad.clone()
    y.grad = None
    mod(x, y, x_pos=pos, y_pos=pos).backward()
    torch.testing.assert_close(g3, 3.0 * y.grad, rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_float64_inputs_are_accepted_like_the_reference():
    """inputs may be float64 (the reference promotes, utils.py:135-142) and are then computed IN float64: float64 GPU
    tensors take the package's torch-op route (one warning), value and gradient come back in float64 and agree with the HIP
    float32 evaluation to float32 accuracy; float64 CPU tensors give the same float64 numbers (ATen CPU vs GPU: ~1e-12)."""
    import warnings
    from sot_amd.losses import Wasserstein1D, wasserstein_1d
    native()
    dev = device()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand(9, 300, device=dev, generator=g, dtype=torch.float64)
    y = torch.rand(9, 300, device=dev, generator=g, dtype=torch.float64).requires_grad_(True)
    pos = torch.linspace(0, 1, 300, device=dev, dtype=torch.float64)
    mod = Wasserstein1D(p=2, square_dist=True).to(dev)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        out = mod(x, y, x_pos=pos, y_pos=pos.clone())
    assert out.dtype == torch.float64 and out.is_cuda
    out.backward()
    assert y.grad is not None and y.grad.dtype == torch.float64 and torch.isfinite(y.grad).all()
    y32 = y.detach().float().requires_grad_(True)
    ref = mod(x.float(), y32, x_pos=pos.float(), y_pos=pos.float().clone())   # the HIP kernels
    ref.backward()
    assert abs(float(out) - float(ref)) <= 2e-5 * abs(float(out))
    assert float((y.grad.float() - y32.grad).abs().max()) <= 2e-3 * float(y32.grad.abs().max())
    yc = y.detach().cpu().requires_grad_(True)
    cpu = mod.cpu()(x.cpu(), yc, x_pos=pos.cpu(), y_pos=pos.cpu().clone())
    mod.to(dev)
    assert abs(float(cpu) - float(out)) <= 1e-10 * abs(float(out))
    rows = wasserstein_1d(pos.expand(9, 300), pos.expand(9, 300), x / x.sum(1, keepdim=True), y.detach() / y.detach().sum(1, keepdim=True))
    assert rows.dtype == torch.float64 and rows.shape == (9,)
    rows_m = mod.row_losses(x, y.detach(), x_pos=pos, y_pos=pos)
    assert rows_m.dtype == torch.float64 and rows_m.shape == (9,)


@pytest.mark.parametrize("case", ["shared_sorted_p1", "shared_unsorted_p2_cutoff", "rows_unsorted_p2", "rows_p3_nm", "positions_only_mean"])
def test_position_gradients_match_the_reference_autograd(case):
    """Gradients w.r.t. the support positions (losses.py:287-313 is differentiable in u_values / v_values through torch.sort and
    take_along_dim; VERDICT round 1, missing #6) against autograd of the op-for-op restatement of the reference on the CPU."""
    from gpu_util import device, native
    from oracle import torch_restatement as tr
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    g = torch.Generator().manual_seed(300 + len(case))   # a fixed seed per case
    B, n, m = 12, 67, 67
    kw, shared, unsorted = dict(p=1), True, False
    if case == "shared_unsorted_p2_cutoff":
        kw, unsorted = dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True), True
    elif case == "rows_unsorted_p2":
        kw, shared, unsorted = dict(p=2), False, True
    elif case == "rows_p3_nm":
        kw, shared, m = dict(p=3), False, 41
    x, y = torch.rand(B, n, generator=g) + 0.05, torch.rand(B, m, generator=g) + 0.05
    def positions(width):
        pos = torch.rand(B if not shared else 1, width, generator=g)
        pos = pos if unsorted else torch.sort(pos, dim=1)[0]
        return pos[0].clone() if shared else pos
    xp, yp = positions(n), positions(m)
    up = torch.rand(B, generator=g)
    ctor = {k: v for k, v in kw.items() if k != "p"}
    # reference (restatement) on the CPU
    xr, yr, xpr, ypr = (t.clone().requires_grad_(True) for t in (x, y, xp, yp))
    if case == "positions_only_mean":
        tr.sot_loss(x, y, xpr, ypr, **kw).backward()
    else:
        (tr.sot_loss(xr, yr, xpr, ypr, reduce=False, **kw) * up).sum().backward()
    # HIP path
    xd, yd, xpd, ypd = (t.to(dev).requires_grad_(True) for t in (x, y, xp, yp))
    mod = Wasserstein1D(p=kw["p"], **ctor).to(dev)
    if case == "positions_only_mean":      # only the positions need a gradient, default reduction (the fused-mean node)
        mod(x.to(dev), y.to(dev), x_pos=xpd, y_pos=ypd).backward()
    else:
        (mod.row_losses(xd, yd, xpd, ypd) * up.to(dev)).sum().backward()
        for got, want in ((xd.grad, xr.grad), (yd.grad, yr.grad)):
            assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    for got, want in ((xpd.grad, xpr.grad), (ypd.grad, ypr.grad)):
        assert got.shape == want.shape
        assert float((got.cpu() - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1e-6), case


@pytest.mark.parametrize("shape", [(5, 67, 67), (7, 700, 700), (6, 1025, 300), (5, 2048, 2048), (3, 5000, 4100), (9, 1, 40), (4, 3, 1)])
@pytest.mark.parametrize("mode", ["p1", "p2_cutoff_dn", "p1.5_sparse", "p3_unsorted"])
@pytest.mark.parametrize("shared", [True, False])
def test_position_gradient_kernel_all_geometries(shape, mode, shared):
    """sot_w1d_position_grad (round 4: the HIP kernel behind gradients w.r.t. the support positions, losses.py:287-298 + 214-220)
    on every thread geometry of the generic kernels (64 / 128 / 256 / 1024 threads per row, 8 / 12 / 16 points per thread), n != m,
    one-point measures, sparse rows (runs of tied levels), un-normalised rows (levels past the other side's last level: the clamp of
    losses.py:220), the cutoff and unsorted positions (the gradient goes back through the sort permutation) -- against autograd of
    the op-for-op restatement on the CPU in float32 (same CDF bits, hence the same ties), entry by entry at 2e-5 of the largest
    entry; and bit-identical between two launches (no atomics)."""
    from oracle import torch_restatement as tr
    from sot_amd import losses as L
    nat = native()
    dev = device()
    B, n, m = shape
    if not shared and n == 5000:
        n, m = 3000, 2500   # per-row positions are sorted on power-of-two arrays: 4096 + 4096 with gradient slots is what one CU holds
    g = torch.Generator().manual_seed(1000 * n + m + len(mode) + int(shared))
    kw = dict(p1=dict(p=1), p2_cutoff_dn=dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True),
              **{"p1.5_sparse": dict(p=1.5), "p3_unsorted": dict(p=3, dont_normalize=True)})[mode]
    x, y = torch.rand(B, n, generator=g) + 0.01, torch.rand(B, m, generator=g) + 0.01
    if mode == "p1.5_sparse":
        x = x * (torch.rand(B, n, generator=g) < 0.3)
        y = y * (torch.rand(B, m, generator=g) < 0.3)
    unsorted = mode == "p3_unsorted"
    def positions(width):   # distinct positions: between EQUAL positions torch.sort's (unstable) order decides which one gets the gradient
        rows = 1 if shared else B
        pos = (torch.arange(width)[None, :] + 0.9 * torch.rand(rows, width, generator=g)) / width
        if unsorted:
            pos = torch.stack([r[torch.randperm(width, generator=g)] for r in pos])
        return pos[0].clone() if shared else pos
    xp, yp = positions(n), positions(m)
    up = torch.rand(B, generator=g) + 0.5
    xpr, ypr = xp.clone().requires_grad_(True), yp.clone().requires_grad_(True)
    (tr.sot_loss(x, y, xpr, ypr, reduce=False, **kw) * up).sum().backward()
    ctor = {k: v for k, v in kw.items() if k != "p"}
    mod = L.Wasserstein1D(p=kw["p"], **ctor).to(dev)
    grads = []
    for _ in range(2):
        xpd, ypd = xp.to(dev).requires_grad_(True), yp.to(dev).requires_grad_(True)
        (mod.row_losses(x.to(dev), y.to(dev), xpd, ypd) * up.to(dev)).sum().backward()
        grads.append((xpd.grad.clone(), ypd.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1]), "position gradients must be deterministic"
    for got, want in ((grads[0][0], xpr.grad), (grads[0][1], ypr.grad)):
        assert got.shape == want.shape
        scale = max(float(want.abs().max()), 1e-6)
        assert float((got.cpu() - want).abs().max()) <= 2e-5 * scale, (shape, mode, shared)


@pytest.mark.gpu
def test_position_gradient_kernel_matches_the_torch_op_route_and_column_sum():
    """The kernel against the package's own torch-op composition of the same gradient (quantile kernel + searchsorted + scatter_add_,
    kept for rows beyond the kernel's LDS budget) on a full-size batch, and sot_column_sum against a float64 torch sum."""
    from sot_amd import losses as L
    nat = native()
    dev = device()
    g = torch.Generator(device=dev).manual_seed(11)
    B, n = 512, 2048
    x, y = torch.rand(B, n, device=dev, generator=g), torch.rand(B, n, device=dev, generator=g)
    pos = torch.sort(torch.rand(B, n, device=dev, generator=g), dim=1)[0]
    pos2 = torch.sort(torch.rand(B, n, device=dev, generator=g), dim=1)[0]
    up = torch.rand(B, device=dev, generator=g)
    flags = L._flags(True, True, True, True)
    a = nat.position_grads(x, y, pos, pos2, 2.0, flags, up)
    b = L._position_grads_torch(x, y, pos, pos2, 2.0, flags, None, up, True, True)
    for got, want in zip(a, b):
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    rows = torch.randn(1000, 1025, device=dev, generator=g)
    out = torch.empty(1025, device=dev)
    lib = nat.load()
    nat.check(lib.sot_column_sum(rows.data_ptr(), 1000, 1025, 1025, out.data_ptr(), nat.stream_ptr(dev)))
    torch.testing.assert_close(out, rows.double().sum(0).float(), rtol=1e-6, atol=1e-6)   # both sums are fp64-accumulated


@pytest.mark.gpu
def test_cpp_host_path_equals_the_python_binding():
    """The module's default call goes through the C++ host path (csrc/sot_torch_glue.cpp: one pybind11 call, C++ autograd node); it
    must give bit for bit what the Python binding gives (same C-ABI calls): value, gradient, an upstream scalar other than 1
    (MixOfLosses weights), a second backward through a retained graph, inference mode, 3-D inputs, strided rows, p = 1 on one grid."""
    import sot_amd.losses as L
    from sot_amd.losses import Wasserstein1D
    nat = native()
    dev = device()
    assert nat.glue() is not None, "the in-tree _sot_glue.so must load on the GPU box"
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand(6, 5, 1025, device=dev, generator=g)
    y = torch.rand(6, 5, 1025, device=dev, generator=g)
    pos = torch.linspace(0, 1, 1025, device=dev)
    pos2 = pos.clone()
    for ctor in (dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True), dict(p=1), dict(p=3)):
        mod = Wasserstein1D(**ctor).to(dev)
        x2, y2, xp, yp, flags, plan, _ = mod._marshal(x, y, pos, pos2, {})
        want_mean, _, want_gy = nat.loss_and_grad(x2, y2, xp, yp, float(mod.p), flags, plan)
        ya = y.clone().requires_grad_(True)
        out = mod(x, ya, x_pos=pos, y_pos=pos2)
        assert out.grad_fn is not None and type(out.grad_fn).__name__ == "CppFunction" and "FusedMeanLoss" in out.grad_fn.name(), \
            "the Python autograd.Function ran instead of the C++ node"
        assert torch.equal(out.detach(), want_mean)
        (out * 0.37).backward(retain_graph=True)
        assert torch.equal(ya.grad.reshape(-1, 1025), want_gy * 0.37)
        ya.grad = None
        out.backward()                                    # second backward: recomputed by the backward kernel
        torch.testing.assert_close(ya.grad.reshape(-1, 1025), want_gy, rtol=1e-6, atol=1e-12)
        with torch.inference_mode():
            assert torch.equal(mod(x, y, x_pos=pos, y_pos=pos2), nat.loss_fused(x2, y2, xp, yp, float(mod.p), flags, plan)[0])
        # the Python binding on the same call
        L.EARLY_GRADIENT = False
        try:
            yb = y.clone().requires_grad_(True)
            ref = mod(x, yb, x_pos=pos, y_pos=pos2)
            assert type(ref.grad_fn).__name__.startswith("_FusedMeanLoss")
            ref.backward()
        finally:
            L.EARLY_GRADIENT = True
        assert torch.equal(ref.detach(), out.detach())
        torch.testing.assert_close(yb.grad, ya.grad, rtol=1e-6, atol=1e-12)
    # strided rows (a slice of a wider buffer) and a weight on x that must NOT take this path
    wide = torch.rand(30, 1100, device=dev, generator=g)
    xv, yv = wide[:, 7:1032], wide[:, 40:1065].clone().requires_grad_(True)
    mod = Wasserstein1D(p=2, square_dist=True).to(dev)
    a = mod(xv, yv, x_pos=pos, y_pos=pos2)
    b = mod(xv.contiguous(), yv, x_pos=pos, y_pos=pos2)
    assert torch.equal(a, b)
    xg = xv.clone().requires_grad_(True)
    both = mod(xg, yv, x_pos=pos, y_pos=pos2)
    assert type(both.grad_fn).__name__.startswith("_FusedMeanLoss")   # both gradients: the Python node with the two-gradient backward kernel
    both.backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


@pytest.mark.gpu
def test_hot_call_cache_is_invalidated_by_everything_it_depends_on():
    """Wasserstein1D remembers its last default call (same position tensors, no keywords) and then goes straight to the C++ host path.
    The entry must not survive: other position tensors, an in-place change of the positions, changed module settings, call keywords,
    other row lengths, a gradient wanted for x, EARLY_GRADIENT off."""
    import sot_amd.losses as L
    from sot_amd.losses import Wasserstein1D
    nat = native()
    dev = device()
    g = torch.Generator(device=dev).manual_seed(12)
    x, y = torch.rand(8, 300, device=dev, generator=g), torch.rand(8, 300, device=dev, generator=g)
    pos = torch.linspace(0, 1, 300, device=dev)
    pos2 = pos.clone()
    mod = Wasserstein1D(p=2, square_dist=True).to(dev)

    def ref(m, a, b, px, py, **kw):
        m2 = Wasserstein1D(p=m.p, square_dist=m.square_dist, dont_normalize=m.dont_normalize, limit_quantile_range=m.limit_quantile_range,
                           require_sort=m.require_sort).to(dev)
        return m2(a, b, x_pos=px, y_pos=py, **kw)

    first = mod(x, y, x_pos=pos, y_pos=pos2)
    assert mod._hot is not None
    assert torch.equal(mod(x, y, x_pos=pos, y_pos=pos2), first)                       # the hot path gives the same bits
    other = torch.sort(torch.rand(300, device=dev, generator=g)).values
    assert torch.equal(mod(x, y, x_pos=other, y_pos=other.clone()), ref(mod, x, y, other, other.clone()))
    mod(x, y, x_pos=pos, y_pos=pos2)
    pos2.mul_(0.5)                                                                     # in-place change: version counter
    assert torch.equal(mod(x, y, x_pos=pos, y_pos=pos2), ref(mod, x, y, pos, pos2.clone()))
    mod.dont_normalize = True                                                          # a module setting changed after the entry was made
    assert torch.equal(mod(x, y, x_pos=pos, y_pos=pos2), ref(mod, x, y, pos, pos2))
    assert torch.equal(mod(x, y, x_pos=pos, y_pos=pos2, limit_quantile_range=True),    # call keyword
                       Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)(x, y, x_pos=pos, y_pos=pos2))
    with pytest.raises(RuntimeError):
        mod(x[:, :200], y[:, :200], x_pos=pos, y_pos=pos2)                            # other row length than the positions
    xg = x.clone().requires_grad_(True)
    out = mod(xg, y, x_pos=pos, y_pos=pos2)
    out.backward()
    assert xg.grad is not None                                                         # gradient for x: not the hot path's case
    L.EARLY_GRADIENT = False
    try:
        yg = y.clone().requires_grad_(True)
        assert type(mod(x, yg, x_pos=pos, y_pos=pos2).grad_fn).__name__.startswith("_FusedMeanLoss")
    finally:
        L.EARLY_GRADIENT = True
    m3 = Wasserstein1D(p=1, fixed_x=300).to(dev)                                        # the fixed_x buffer form, 3-D rows
    a = m3(x.reshape(2, 4, 300), y.reshape(2, 4, 300))
    assert torch.equal(m3(x.reshape(2, 4, 300), y.reshape(2, 4, 300)), a) and m3._hot is not None
    assert torch.equal(a, m3(x, y))


def test_module_under_make_graphed_callables():
    """torch.cuda.make_graphed_callables(module, ...) -- forward and backward each captured into a HIP graph by PyTorch -- gives the eager
    module's loss and gradient bit for bit, also on inputs other than the ones it was captured with."""
    import warnings
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    B, N = 256, 1025
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True, fixed_x=N).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    sample_x = torch.rand(B, N, device=dev, generator=g)
    sample_y = torch.rand(B, N, device=dev, generator=g).requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        graphed = torch.cuda.make_graphed_callables(mod, (sample_x, sample_y))
        for _ in range(2):
            x = torch.rand(B, N, device=dev, generator=g)
            y = torch.rand(B, N, device=dev, generator=g).requires_grad_(True)
            want = mod(x, y)
            want.backward()
            y2 = y.detach().clone().requires_grad_(True)
            got = graphed(x, y2)
            got.backward()
            torch.cuda.synchronize()
            assert torch.equal(got.detach(), want.detach())
            assert torch.equal(y2.grad, y.grad)


def test_user_graph_of_the_trainers_fresh_position_step():
    """INTEGRATION.md's HIP-graph recipe on the reference trainer's call pattern (trainer.py:187-228: x_pos rebuilt and y_pos = x_pos.clone()
    on every step): `loss = mod(x, y, x_pos=, y_pos=); loss.backward()` captured once with the position arithmetic INSIDE the capture (the
    position plan's launch becomes part of the graph), replayed on new data copied into the static tensors -- loss and gradient bit for bit
    those of the eager module; the same through a wrapper module under torch.cuda.make_graphed_callables."""
    import warnings
    from sot_amd import spectra
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    B, N = 1024, 1025
    mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
    freqs = torch.fft.rfftfreq(2048, d=1.0 / 16000.0).to(dev)
    g = torch.Generator(device=dev).manual_seed(9)
    xs = torch.rand(B, N, device=dev, generator=g)
    ys = torch.rand(B, N, device=dev, generator=g).requires_grad_(True)

    def step():
        x_pos = freqs / freqs.max()
        return mod(xs, ys, x_pos=x_pos, y_pos=x_pos.clone())

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ys.grad = None
            step().backward()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    ys.grad = None
    with torch.cuda.graph(graph):
        loss = step()
        loss.backward()
    pos = spectra.unit_frequencies(2048, 16000.0, dev)
    for _ in range(3):
        nx, ny = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
        with torch.no_grad():
            xs.copy_(nx)
            ys.copy_(ny)
        graph.replay()
        yc = ny.clone().requires_grad_(True)
        want = mod(nx, yc, x_pos=pos, y_pos=pos.clone())
        want.backward()
        torch.cuda.synchronize()
        assert torch.equal(loss.detach(), want.detach())
        assert torch.equal(ys.grad, yc.grad)

    class Step(torch.nn.Module):          # make_graphed_callables wants a module / function of tensors
        def __init__(self):
            super().__init__()
            self.loss = mod

        def forward(self, x, y, f):
            x_pos = f / f.max()
            return self.loss(x, y, x_pos=x_pos, y_pos=x_pos.clone())

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        graphed = torch.cuda.make_graphed_callables(Step().to(dev), (xs.detach().clone(), ys.detach().clone().requires_grad_(True), freqs.clone()))
        nx, ny = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
        y1, y2 = ny.clone().requires_grad_(True), ny.clone().requires_grad_(True)
        want = mod(nx, y1, x_pos=pos, y_pos=pos.clone())
        want.backward()
        got = graphed(nx, y2, freqs)
        got.backward()
        torch.cuda.synchronize()
        assert torch.equal(got.detach(), want.detach()) and torch.equal(y2.grad, y1.grad)


def test_hot_call_cache_does_not_swallow_position_gradients():
    """an entry made under no_grad (a validation step) must not serve a later training call whose POSITIONS require a
    gradient -- requires_grad_() does not bump the version counter, so identity + version alone would still match."""
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    g = torch.Generator(device=dev).manual_seed(3)
    x, y = torch.rand(6, 200, device=dev, generator=g), torch.rand(6, 200, device=dev, generator=g)
    pos = torch.sort(torch.rand(200, device=dev, generator=g)).values
    pos2 = pos.clone()
    mod = Wasserstein1D(p=2).to(dev)
    with torch.no_grad():
        first = mod(x, y, x_pos=pos, y_pos=pos2)
    assert mod._hot is not None
    pos.requires_grad_(True)
    pos2.requires_grad_(True)
    out = mod(x, y, x_pos=pos, y_pos=pos2)
    assert torch.equal(out.detach(), first)
    out.backward()
    assert pos.grad is not None and pos2.grad is not None and float(pos.grad.abs().sum()) > 0
    ref = Wasserstein1D(p=2)   # the same module on CPU tensors (torch-op route): autograd through sort / take_along_dim
    pc, pc2 = pos.detach().cpu().requires_grad_(True), pos2.detach().cpu().requires_grad_(True)
    ref(x.cpu(), y.cpu(), x_pos=pc, y_pos=pc2).backward()
    torch.testing.assert_close(pos.grad.cpu(), pc.grad, rtol=2e-4, atol=1e-7)
    torch.testing.assert_close(pos2.grad.cpu(), pc2.grad, rtol=2e-4, atol=1e-7)


def test_one_long_side_takes_the_torch_route_instead_of_raising():
    """n + m within the LDS limit but ONE side beyond the kernels' 16384-point cap: the promised torch-op route, not
    SOT_ERR_UNSUPPORTED_SIZE."""
    import warnings
    from sot_amd.losses import Wasserstein1D
    native()
    dev = device()
    g = torch.Generator(device=dev).manual_seed(4)
    x, y = torch.rand(2, 17000, device=dev, generator=g), torch.rand(2, 500, device=dev, generator=g)
    px, py = torch.linspace(0, 1, 17000, device=dev), torch.linspace(0, 1, 500, device=dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = Wasserstein1D(p=1)(x, y, x_pos=px, y_pos=py)
    want = Wasserstein1D(p=1)(x.cpu(), y.cpu(), x_pos=px.cpu(), y_pos=py.cpu())
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=0)
