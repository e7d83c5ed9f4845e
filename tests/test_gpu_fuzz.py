"""-m gpu: a fixed slice of tools/fuzz_gpu.py's randomised sweep -- random lengths (around every geometry switch and arbitrary), position
kinds (shared / per row, sorted / unsorted / tied), weight kinds (dyadic, sparse, degenerate rows), modes, p, strides and plans --
through the C ABI against the C oracle: forward rows, both gradients, and the training form against forward + backward."""
import os
import sys

import pytest

from conftest import ROOT
from gpu_util import native

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_cases_against_the_oracle(seed):
    native()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    # gradients: 2e-4 of the row's gradient scale -- rows of a handful of points against thousands under dont_normalize sit at
    # 1e-4 (both sides within 1e-6 of float64 autograd; tools/fuzz_repro.py); everything else stays below 2e-5
    cases, failures, worst_forward, worst_grad = fuzz_gpu.run(budget=120.0, seed0=seed, max_cases=400, grad_tol=2e-4, verbose=False)
    assert cases == 400
    assert failures == [], failures[:3]
    assert worst_forward <= 1e-5


@pytest.mark.parametrize("seed", [21, 22])
def test_random_stft_cases_against_torch_stft(seed):
    """tools/fuzz_stft.py: random n_fft (64 ... 4096), hop, clip length, batch, window and signal kind; magnitudes against torch.stft,
    gradients against its autograd, and the backward from the stored spectrum against the recomputing one (bit-identical)."""
    native()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_stft
    cases, failures, worst_forward, worst_grad = fuzz_stft.run(budget=120.0, seed0=seed, max_cases=300, verbose=False)
    assert cases == 300
    # a failure record is (kind, case, forward error, gradient error, stored == recomputed); the gradient of |X| is ill-conditioned
    # where |X| ~ 0 (pure tones between bins): allow those up to 2e-2 of the largest gradient entry, nothing else
    hard = [f for f in failures if f[0] != "STFT" or f[2] > 2e-5 or f[3] > 2e-2 or not f[4]]
    assert hard == [], hard[:3]
    assert worst_forward <= 2e-5


@pytest.mark.parametrize("seed", [31, 32])
def test_random_module_calls_against_the_cpu_route(seed):
    """tools/fuzz_module.py: the drop-in module on GPU tensors (C++ host path, hot-call cache, Python binding, row losses, `dims`, hinge, 3-D
    inputs, fixed_x / explicit / unsorted / per-row positions, every mode, x / y / both gradients, three calls per module) against the same
    module on CPU tensors (the torch-op route, bit-identical to the reference)."""
    native()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_module
    # losses to 3e-5; gradients to 2 % of the row's largest entry: float32 CDF levels of the two measures tie exactly now and then (N^2 / 2^24
    # per row), and at a tie the gradient is a convention -- a wrong scale, sign or routing would be an O(1) error
    cases, failures, worst_loss, worst_grad = fuzz_module.run(budget=120.0, seed0=seed, max_cases=150, verbose=False, grad_tol=2e-2)
    assert cases == 150
    assert failures == [], failures[:3]


@pytest.mark.parametrize("seed", [41, 42])
def test_random_mss_cases_against_float64(seed):
    """tools/r5/fuzz_mss.py (round-5 review: it existed beside the suite, not in it): random clip length (1 ... 20 000 samples, around every frame
    boundary), batch, subset of the six transform sizes, L1 / L2, magnitude / log-magnitude weights, per-clip means -- MSSLoss on GPU tensors
    against the reference's op sequence in float64, with the reference's own float32 error as the yardstick (loss within 4 x + 1e-5; gradient norm
    within 4 x + 5e-6, or -- single bins changing the sign of |T| - |V| / crossing safe_log's threshold, a lottery every float32 chain plays --
    the median error within 2 x)."""
    native()
    sys.path.insert(0, os.path.join(ROOT, "tools", "r5"))
    import fuzz_mss
    cases, bad, failures = fuzz_mss.run(budget=170.0, seed0=seed, verbose=False, max_cases=150)
    assert cases == 150
    assert bad == 0, failures[:3]


@pytest.mark.parametrize("seed", [51, 52])
def test_random_per_row_position_cases_on_every_route(seed):
    """tools/r6/fuzz_rowpos.py (round 6): per-row positions -- 2048-point rows (the compile-time RP kernels) most of the time, every other length on the
    generic gathering kernels -- with random, sorted, descending, clustered, tied, one-interval and duplicated positions mixed row by row, random weights,
    modes, p, strides: the default route (pre-sort kernel), SOT_FLAG_NO_SPECIALIZE (the row kernel's own merge sort) and the hand-over of stored
    permutations agree BIT FOR BIT on forward rows, both weight gradients and the position gradients; the stored permutations are the stable argsort;
    forward rows within 2e-5 of the C oracle, weight gradients within 2e-4 (p = 1, 2; 1e-3 general p) of the row's gradient scale."""
    native()
    sys.path.insert(0, os.path.join(ROOT, "tools", "r6"))
    import fuzz_rowpos
    cases, failures, worst_forward, worst_grad = fuzz_rowpos.run(budget=150.0, seed0=seed, max_cases=250, verbose=False)
    assert cases == 250
    assert failures == [], failures[:3]
    assert worst_forward <= 2e-5
