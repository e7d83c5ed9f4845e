"""CPU tests: the oracle (oracle/sot_oracle.c + numpy restatement) against the golden vectors
captured from the imported reference (oracle/make_golden.py), and against torch's ATen kernels."""
import numpy as np
import pytest
import torch

from conftest import case_names, ctor_to_flags, load_case
from oracle import sot_oracle as so


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("n", list(range(1, 70)) + [127, 128, 129, 255, 256, 257, 511, 512, 513, 1025, 2048, 2049,
                                                     4096, 4100, 5000, 8191, 16384, 16400, 20000])
def test_aten_sum_order_matches_torch(n):
    rng = np.random.default_rng(n)
    x = (rng.random((5, n)) ** int(rng.integers(1, 9))).astype(np.float32)
    want = torch.sum(torch.from_numpy(x), dim=1, keepdim=True).numpy()[:, 0]
    got_c = np.array([so.aten_sum(r) for r in x], np.float32)
    got_np = so.aten_sum_rows_np(x)
    assert (bits(got_c) == bits(want)).all()
    assert (bits(got_np) == bits(want)).all()


@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_reference_fixture(name, manifest):
    meta, g = load_case(name, manifest)
    p, flags = ctor_to_flags(meta["ctor"])
    x = g["x"].reshape(-1, g["x"].shape[-1])
    y = g["y"].reshape(-1, g["y"].shape[-1])
    rows, d = so.forward(x, y, g["x_pos"], g["y_pos"], p=p, flags=flags, debug=True)
    # every stored intermediate is bit-identical to the reference's
    for key in ("U", "V", "Q", "uq", "vq"):
        if key in g:
            assert (bits(d[key]) == bits(g[key].reshape(d[key].shape))).all(), key
    if "U_last" in g:
        assert (bits(d["U"][:, -1]) == bits(g["U_last"])).all()
        assert (bits(d["V"][:, -1]) == bits(g["V_last"])).all()
    if "x_sorter" in g:
        assert (d["xsorter"] == g["x_sorter"]).all() and (d["ysorter"] == g["y_sorter"]).all()
    want_rows = g["row_loss"].reshape(-1)
    if p in (1.0, 2.0):
        assert (bits(rows) == bits(want_rows)).all()
        assert bits(so.mean(rows)) == bits(g["scalar"])
    else:  # powf (libm) vs torch's vectorised pow: last-bit differences allowed
        np.testing.assert_allclose(rows, want_rows, rtol=2e-6, atol=0)


@pytest.mark.parametrize("name", [n for n in case_names() if n.startswith(("b4n512", "nm_", "unsorted", "edge"))])
def test_numpy_restatement_agrees_with_c_oracle(name, manifest):
    meta, g = load_case(name, manifest)
    p, flags = ctor_to_flags(meta["ctor"])
    rows_c = so.forward(g["x"], g["y"], g["x_pos"], g["y_pos"], p=p, flags=flags)
    rows_np = so.forward_numpy(g["x"], g["y"], g["x_pos"], g["y_pos"], p=p, flags=flags)
    if p in (1.0, 2.0):
        assert (bits(rows_c) == bits(rows_np)).all()
    else:
        np.testing.assert_allclose(rows_c, rows_np, rtol=2e-6)


def smooth_rows(d):
    """Rows on which the loss is differentiable: no exact tie between merged levels other than the
    common end level.  At a tie the reference's gradient is decided by its unstable sort's tie order
    (a kink: any order gives a valid subgradient), so only tie-free rows are compared element-wise."""
    ok = []
    for U, V in zip(d["U"], d["V"]):
        top = max(U[-1], V[-1])
        tie_uv = np.intersect1d(U[U < top], V[V < top]).size > 0
        dup = (np.diff(U) == 0).any() or (np.diff(V) == 0).any()
        ok.append(not tie_uv and not dup)
    return np.array(ok)


@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("seeded_b256")])
def test_oracle_backward_matches_reference_autograd(name, manifest):
    """(a) every row: the closed-form backward == autograd through the op-for-op restatement with a
    STABLE level sort (our documented tie convention); (b) tie-free rows: == the gradients the imported
    reference produced (its unstable sort only changes which member of a tie run gets the gradient)."""
    from oracle import torch_restatement as tr
    meta, g = load_case(name, manifest)
    if "grad_x" not in g:
        pytest.skip("no gradients stored")
    c = meta["ctor"]
    p, flags = ctor_to_flags(c)
    x = g["x"].reshape(-1, g["x"].shape[-1])
    y = g["y"].reshape(-1, g["y"].shape[-1])
    xp, yp = g["x_pos"], g["y_pos"]
    xp = xp.reshape(-1, xp.shape[-1]) if xp.ndim == 3 else xp
    yp = yp.reshape(-1, yp.shape[-1]) if yp.ndim == 3 else yp
    B = x.shape[0]
    _, d = so.forward(x, y, xp, yp, p=p, flags=flags, debug=True)
    rows_ok = smooth_rows(d)
    gx, gy = so.backward(x, y, xp, yp, np.full(B, 1.0 / B, np.float32), p=p, flags=flags)
    xt = torch.tensor(x, requires_grad=True)
    yt = torch.tensor(y, requires_grad=True)
    loss = tr.sot_loss(xt, yt, torch.tensor(xp), torch.tensor(yp), p=c.get("p", 1),
                       square_dist=c.get("square_dist", False), dont_normalize=c.get("dont_normalize", False),
                       limit_quantile_range=c.get("limit_quantile_range", False), stable_levels=True)
    sx, sy = torch.autograd.grad(loss, [xt, yt])
    for got, stable, ref in ((gx, sx.numpy(), g["grad_x"].reshape(gx.shape)),
                             (gy, sy.numpy(), g["grad_y"].reshape(gy.shape))):
        tol = 2e-6 * np.abs(stable).max(axis=1, keepdims=True) + 3e-8  # 3e-8: fp32 cancellation noise of autograd on zero-gradient rows
        assert (np.abs(got - stable) <= tol).all(), name
        assert (np.abs(got - ref) <= tol)[rows_ok].all(), name


def test_known_answers():
    n = 64
    pos = np.linspace(0, 1, n, dtype=np.float32)
    f = so.make_flags()
    # identical inputs -> exactly 0
    x = np.random.default_rng(0).random((3, n)).astype(np.float32)
    assert (so.forward(x, x, pos, pos, p=1, flags=f) == 0).all()
    assert (so.forward(x, x, pos, pos, p=2, flags=f) == 0).all()
    # two Diracs: |pos_i - pos_j|^p
    a = np.zeros((1, n), np.float32)
    b = np.zeros((1, n), np.float32)
    a[0, 10] = 3.0
    b[0, 37] = 0.5
    d = np.float32(abs(pos[10] - pos[37]))
    assert so.forward(a, b, pos, pos, p=1, flags=f)[0] == d
    assert so.forward(a, b, pos, pos, p=2, flags=f)[0] == d * d
    # shift by k bins: k/(n-1)
    k = 5
    xs = np.zeros((1, n), np.float32)
    xs[0, 8:20] = np.random.default_rng(1).random(12)
    ys = np.roll(xs, k, axis=1)
    np.testing.assert_allclose(so.forward(xs, ys, pos, pos, p=1, flags=f)[0], k / (n - 1), rtol=1e-5)
    # p < 1 is rejected like losses.py:271
    with pytest.raises(AssertionError):
        so.forward(xs, ys, pos, pos, p=0.5, flags=f)


def test_scipy_cross_check_unsorted_positions():
    from scipy.stats import wasserstein_distance
    rng = np.random.default_rng(3)
    x = rng.random((4, 50)).astype(np.float32)
    y = rng.random((4, 70)).astype(np.float32)
    xp = rng.random((4, 50)).astype(np.float32)
    yp = rng.random((4, 70)).astype(np.float32)
    got = so.forward(x, y, xp, yp, p=1, flags=so.make_flags())
    want = [wasserstein_distance(xp[r], yp[r], x[r], y[r]) for r in range(4)]
    np.testing.assert_allclose(got, want, rtol=5e-6)


def test_oracle_matches_the_reference_on_the_dyadic_full_size_fixtures():
    """tests/golden/dyadic_full_size.npz (oracle/make_golden_dyadic.py: the reference on dyadic weights, SURVEY B.1 iii): the C
    oracle reproduces the reference's row losses bit for bit on the masked-dense config-4 rows and on the config-5 SOT stage
    (a strided sample of rows each: the oracle is a single-threaded checker), and its result does not depend on whether the
    zero-weight points of a ragged row are present (the CSR form) -- exactly, which is what makes these fixtures a fair
    100 %-of-rows test for the CSR kernel in cutoff mode."""
    import os
    from conftest import GOLDEN
    from oracle.inputs import sha256_of
    from sot_amd.bench_inputs import dyadic_pairs, dyadic_ragged_supports
    fx = np.load(os.path.join(GOLDEN, "dyadic_full_size.npz"))
    rs = dyadic_ragged_supports(8192, 512, int(fx["seed"]))
    xm, ym = rs["dense"]
    assert sha256_of(xm, ym) == bytes(fx["c4_inputs_sha256"]).hex()
    pos = rs["pos"].numpy()
    (xw, xp, xo), (yw, yp, yo) = rs["csr"]
    sel = np.arange(0, 8192, 37)
    for mode, (p, flags) in (("cutoff", (2.0, so.make_flags(True, True, True, True))), ("p1", (1.0, so.make_flags()))):
        rows = so.forward(xm.numpy()[sel], ym.numpy()[sel], pos, pos, p=p, flags=flags)
        assert (bits(rows) == bits(fx[f"c4_{mode}_rows"][sel])).all(), mode
        for r in sel[:40]:
            a, b, e, f = int(xo[r]), int(xo[r + 1]), int(yo[r]), int(yo[r + 1])
            want = fx[f"c4_{mode}_rows"][r]
            ragged = so.forward(xw[a:b].numpy()[None], yw[e:f].numpy()[None], xp[a:b].numpy(), yp[e:f].numpy(), p=p, flags=flags)[0]
            # the same supports with the grid's LAST point kept as a zero-weight point on both sides
            xr, xq = np.append(xw[a:b].numpy(), np.float32(0)), np.append(xp[a:b].numpy(), pos[-1])
            yr, yq = np.append(yw[e:f].numpy(), np.float32(0)), np.append(yp[e:f].numpy(), pos[-1])
            anchored = so.forward(xr[None], yr[None], xq, yq, p=p, flags=flags)[0]
            assert abs(anchored - want) <= 2e-6 * abs(want), (mode, r, anchored, want)   # same levels, other summation order of the terms
            if mode == "p1":   # both CDFs end at the same mass: no level ever clamps to the last point, removed points are inert
                assert abs(ragged - want) <= 2e-6 * abs(want), (mode, r, ragged, want)
    # (dont_normalize: when y carries less mass than x, the levels between V_last and U_last take ys[m - 1] -- losses.py:220's clamp --
    #  which is the grid's last point in the masked-dense form and the last KEPT point in the ragged form: there the two forms are
    #  different problems by the reference's own semantics, not a rounding matter; `anchored` above is the ragged form of the dense one.)
    x, y = dyadic_pairs(4096, 1025, int(fx["seed"]) + 1)
    assert sha256_of(x, y) == bytes(fx["c5_inputs_sha256"]).hex()
    f = torch.fft.rfftfreq(2048, d=1.0 / 16000.0)
    p5 = (f / f.max()).float().numpy()
    sel = np.arange(0, 4096, 29)
    rows = so.forward(x.numpy()[sel], y.numpy()[sel], p5, p5, p=2.0, flags=so.make_flags(True, True, True, True))
    assert (bits(rows) == bits(fx["c5_rows"][sel])).all()
