"""CPU tests: host-side mirror of the reference interface, the C-ABI library's exports, and error
behaviour that does not need a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def test_library_builds_and_exports_every_declared_symbol():
    import sot_amd
    lib_path = sot_amd.build.build()
    assert os.path.exists(lib_path)
    handle = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, "include", "sot_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|size_t|const char \*)\s*\*?\s*(sot_\w+)\s*\(", header, flags=re.M))
    assert {"sot_w1d_forward", "sot_w1d_backward", "sot_w1d_reduce_mean", "sot_w1d_quantiles", "sot_segmented_sort",
            "sot_prepare_positions", "sot_workspace_bytes", "sot_abi_version", "sot_status_string", "sot_stft_frames",
            "sot_stft_mag_forward", "sot_stft_mag_backward"} <= declared
    for name in declared:
        assert hasattr(handle, name), name
    assert set(sot_amd._native.EXPORTS) == declared
    handle.sot_abi_version.restype = ctypes.c_int
    m = re.search(r"#define SOT_ABI_VERSION (\d+)", header)
    assert handle.sot_abi_version() == int(m.group(1)) == sot_amd._native.ABI_VERSION   # header, library and binding agree
    handle.sot_status_string.restype = ctypes.c_char_p
    assert b"p>=1" in handle.sot_status_string(-1)


def test_host_side_validation_without_gpu():
    """C-ABI argument validation runs on the host before anything is enqueued."""
    from sot_amd import _native as nat
    lib = nat.load()
    pr = nat.SotProblem()
    pr.B, pr.n, pr.m = 4, 16, 16
    pr.x_row_stride = pr.y_row_stride = 16
    pr.p, pr.flags = 0.5, 0
    assert lib.sot_w1d_forward(ctypes.byref(pr), None, None, 0, None) == -4  # row_loss NULL
    buf = (ctypes.c_float * 64)()
    addr = ctypes.addressof(buf)
    assert lib.sot_w1d_forward(ctypes.byref(pr), addr, None, 0, None) == nat.SOT_ERR_INVALID_P
    pr.p = 1.0
    assert lib.sot_w1d_forward(ctypes.byref(pr), addr, None, 0, None) == -4  # x/y NULL
    pr.n = 0
    assert lib.sot_w1d_forward(ctypes.byref(pr), addr, None, 0, None) == -2
    pr.n = 16
    assert lib.sot_workspace_bytes(ctypes.byref(pr)) >= 2 * 16 * 8 + 8
    with pytest.raises(AssertionError, match="only valid for p>=1"):
        nat.check(nat.SOT_ERR_INVALID_P, 0.5)
    with pytest.raises(nat.SotError):
        nat.check(-3)


def test_module_surface_matches_reference():
    from sot_amd.losses import MixOfLosses, Wasserstein1D
    # paper YAML init_args (SOT-2048/c6m8cytv-42/train_config.yaml:89-99), incl. the ignored cumsum_only
    m = Wasserstein1D(p=2, fixed_x=None, require_sort=True, log_scaled_x=False, cumsum_only=False, hinge=False,
                      square_dist=True, dont_normalize=True, limit_quantile_range=True)
    assert m.__class__.__name__ == "Wasserstein1D"           # trainer.py:216, losses.py:361
    assert (m.p, m.require_sort, m.log_scaled_x) == (2, True, False)
    assert (m.dont_normalize, m.limit_quantile_range, m.hinge, m.square_dist) == (True, True, False, True)
    assert m.fixed_x is None and list(m.state_dict()) == []
    m2 = Wasserstein1D(p=1, fixed_x=257)
    assert torch.equal(m2.fixed_x, torch.linspace(0, 1, 257)) and list(m2.state_dict()) == ["fixed_x"]
    assert hasattr(m2, "log_scaled_x")                       # trainer.py:187
    with pytest.raises(ValueError, match="x_pos and y_pos must be provided"):
        m(torch.rand(2, 8), torch.rand(2, 8))
    with pytest.raises(AssertionError, match="only valid for p>=1"):
        Wasserstein1D(p=0.5, fixed_x=8)(torch.rand(2, 8), torch.rand(2, 8))
    # CPU tensors take the package's torch-op route (the reference is device-agnostic, losses.py:129-211): tests/test_cpu_path.py;
    # the NATIVE layer has no CPU path and says so
    assert m2(torch.rand(2, 257), torch.rand(2, 257)).ndim == 0
    with pytest.raises(RuntimeError, match="no CPU path"):
        from sot_amd import _native as nat
        nat.require_hip(torch.rand(3))
    with pytest.raises(TypeError, match="float32"):
        from sot_amd import _native as nat
        nat.require_hip(_FakeCuda())  # fp64 "GPU tensor": the dtype check fires without a GPU
    mix = MixOfLosses([m2], [1.0])
    assert mix.losses[0] is m2 and mix.weights == [1.0]


class _FakeCuda:
    """minimal stand-in to exercise the dtype check without a GPU"""
    is_cuda = True
    dtype = torch.float64
    device = "cuda:0"


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "1d-spectral-optimal-transport_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the C oracle", "").replace("CPU oracle", "") or f == "never", (f,)


def test_shard_rows_partition():
    from sot_amd.distributed import shard_rows
    for total in (0, 1, 7, 8192, 65536, 65539):
        for ws in (1, 2, 3, 8):
            blocks = [shard_rows(total, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_bench_inputs_are_the_fixture_generator():
    """bench.py draws its timed inputs from the package (sot_amd.bench_inputs), not from oracle/: the tensors must be the
    ones the golden scalars (tests/golden/manifest.json: _config2_*, _config4_*) were computed on."""
    import torch
    from oracle.inputs import gen_inputs
    from sot_amd.bench_inputs import ragged_supports, spectrum_pairs
    for kind in ("uniform", "peaky"):
        a, b = gen_inputs(kind, 5, 67, 61, 1234), spectrum_pairs(kind, 5, 67, 61, 1234)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    rs = ragged_supports(32, 64, 9)
    xm, ym = rs["dense"]
    (xw, xp, xo), (yw, yp, yo) = rs["csr"]
    assert int(xo[-1]) == xw.numel() == int((xm != 0).sum()) and int(yo[-1]) == yw.numel() == int((ym != 0).sum())
    r = 7
    assert torch.equal(xw[xo[r]:xo[r + 1]], xm[r][xm[r] != 0]) and torch.equal(xp[xo[r]:xo[r + 1]], rs["pos"][xm[r] != 0])


def test_stale_library_is_detected_by_content_not_by_mtime(tmp_path, monkeypatch):
    """_native.load() refuses (or rebuilds) a libsot_hip.so that was built from other sources than the tree's: the digest
    next to the library is a content hash, so a copy of the tree that resets modification times changes nothing."""
    import sot_amd
    b = sot_amd.build
    b.build()
    assert not b.is_stale()
    os.utime(b.DEPS[0])                        # newer mtime, same content: still current
    assert not b.is_stale()
    monkeypatch.setattr(b, "DIGEST", str(tmp_path / "other.digest"))
    assert b.is_stale()                         # no digest -> stale
    (tmp_path / "other.digest").write_text("0" * 64 + "\n")
    assert b.is_stale()                         # digest of other sources -> stale
    (tmp_path / "other.digest").write_text(b.source_digest() + "\n")
    assert not b.is_stale()


def test_cpp_host_path_is_built_in_tree_and_loads_without_a_gpu():
    """_sot_glue.so (build.build_glue: csrc/sot_torch_glue.cpp, no device code) sits next to the package, is current with its
    sources and exports the two entry points the module uses; binding it resolves every C-ABI symbol it calls."""
    import sot_amd
    from sot_amd import _native as nat
    assert os.path.exists(sot_amd.build.GLUE_LIB) and not sot_amd.build.glue_is_stale()
    g = nat.glue()
    assert g is not None and callable(g.mean_loss) and callable(g.bind)
    assert g.bind(nat.library_path()) == nat.ABI_VERSION
    import torch
    with pytest.raises(RuntimeError, match="2-D float32 GPU tensor"):
        t = torch.rand(2, 8)
        i = torch.zeros(8, dtype=torch.int32)
        g.mean_loss(t, t, t[0], t[0], i, i, i[:2], 1.0, 8)


def test_module_can_be_copied_and_pickled():
    """copy.deepcopy (EMA copies) and pickle (torch.save(model)) of a module that has been used: its caches (position plans with
    device events, the hot-call entry with the extension module) are not part of the copy."""
    import copy
    import pickle
    import torch
    from sot_amd.losses import Wasserstein1D
    m = Wasserstein1D(p=2, fixed_x=33, square_dist=True)
    x, y = torch.rand(3, 33), torch.rand(3, 33)
    want = m(x, y)
    m._hot = ("not", "picklable", lambda: None)          # what a GPU call would have left behind
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert clone._hot is None and clone._plans is not None and clone._plans.entries == []
        assert torch.equal(clone.fixed_x, m.fixed_x) and (clone.p, clone.square_dist) == (2, True)
        assert torch.equal(clone(x, y), want)


def test_sibling_losses_of_a_mix_configuration():
    """MeanDifference / KL (losses.py:7-86), the operands a MixOfLosses configuration may name next to Wasserstein1D: known answers."""
    import numpy as np
    import pytest
    import torch
    from sot_amd.losses import KL, MeanDifference, MixOfLosses, Wasserstein1D, mean_difference
    g = torch.Generator().manual_seed(4)
    x, y = torch.rand(2, 3, 5, generator=g), torch.rand(2, 3, 5, generator=g)
    w = torch.rand(5, generator=g)
    xn, yn, wn = x.numpy().astype(np.float64), y.numpy().astype(np.float64), w.numpy().astype(np.float64)
    np.testing.assert_allclose(mean_difference(x, y).item(), np.abs(xn - yn).mean(), rtol=1e-6)
    np.testing.assert_allclose(mean_difference(x, y, "l2", weights=w).item(), ((xn - yn) ** 2 * wn).mean(), rtol=1e-6)
    np.testing.assert_allclose(mean_difference(x, y, "L1", dims=[1, 2]).numpy(), np.abs(xn - yn).mean(axis=(1, 2)), rtol=1e-6)
    with pytest.raises(ValueError):
        mean_difference(x, y, "cosine")
    srt = MeanDifference("L2")(x, y, sort=True)
    np.testing.assert_allclose(srt.item(), ((np.sort(xn, -1) - np.sort(yn, -1)) ** 2).mean(), rtol=1e-6)
    a, b = xn / xn.sum(-1, keepdims=True), yn / yn.sum(-1, keepdims=True)
    kl = (a * (np.log(a + 1e-10) - np.log(b + 1e-10))).sum(-1)
    np.testing.assert_allclose(KL()(x, y).item(), kl.mean(), rtol=1e-5)
    np.testing.assert_allclose(KL(reverse=True)(y, x).item(), kl.mean(), rtol=1e-5)
    np.testing.assert_allclose(KL()(x, y, dims=1).numpy(), kl.mean(axis=1), rtol=1e-5)
    assert KL()(torch.zeros(2, 4), torch.rand(2, 4)).item() == 0.0            # safe_divide: a zero row stays zero
    mix = MixOfLosses([Wasserstein1D(p=1, fixed_x=5), KL(), MeanDifference()], weights=[1.0, 0.5, 2.0])
    out = mix(x, y)
    assert set(out) == {"Wasserstein1D", "KL", "MeanDifference"}
    np.testing.assert_allclose(out["KL"].item(), 0.5 * kl.mean(), rtol=1e-5)
