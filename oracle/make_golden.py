"""oracle/make_golden.py -- generates tests/golden/*.npz FROM THE REFERENCE ITSELF.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  It imports the reference's losses.py behind a 3-module import
shim (nnAudio/librosa are imported by features.py but are
unused by the hot path), evaluates `losses.Wasserstein1D` / `wasserstein_1d`
on seeded inputs, and stores inputs + the reference's outputs and
intermediates as small fixtures.  While doing so it asserts that
oracle/torch_restatement.py is bit-identical to the reference.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import torch_restatement as tr  # noqa: E402
from oracle.inputs import gen_inputs, sha256_of as sha  # noqa: E402


def import_reference():
    for name in ("nnAudio", "nnAudio.features", "librosa"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["nnAudio"].features = sys.modules["nnAudio.features"]
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningDataModule = type("LightningDataModule", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["pytorch_lightning"] = pl
    sys.path.insert(0, "/root/reference")
    import losses  # type: ignore
    import features  # type: ignore
    import synths  # type: ignore
    return losses, features, synths


MODES = {
    # name: (ctor kwargs)  -- SURVEY §8a mode matrix
    "p1": dict(p=1),
    "p2": dict(p=2),
    "cutoff": dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True),
    "nocut": dict(p=2, square_dist=True, dont_normalize=False, limit_quantile_range=False),
    "p3": dict(p=3),
    "p1_cut": dict(p=1, dont_normalize=True, limit_quantile_range=True),
}


def run_case(losses, name, x, y, x_pos, y_pos, ctor, call_kwargs=None, store_inputs=True, store_mid=True,
             grads=True, lead_shape=None):
    """Evaluate the reference; return dict of arrays for the fixture."""
    call_kwargs = call_kwargs or {}
    mod = losses.Wasserstein1D(**ctor)
    xin = x.clone().requires_grad_(grads)
    yin = y.clone().requires_grad_(grads)
    pos_kw = {} if x_pos is None else dict(x_pos=x_pos, y_pos=y_pos)
    scalar = mod(xin, yin, **pos_kw, **call_kwargs)
    out = {"scalar": scalar.detach().numpy()}
    if grads:
        gx, gy = torch.autograd.grad(scalar, [xin, yin])
        out["grad_x"], out["grad_y"] = gx.numpy(), gy.numpy()
    with torch.no_grad():
        # row losses: same module, dims chosen so that nothing is averaged
        x2 = x.reshape(-1, x.shape[-1])
        y2 = y.reshape(-1, y.shape[-1])
        pk = {}
        if x_pos is not None:
            pk = dict(x_pos=x_pos.reshape(-1, x_pos.shape[-1]) if x_pos.ndim == 3 else x_pos,
                      y_pos=y_pos.reshape(-1, y_pos.shape[-1]) if y_pos.ndim == 3 else y_pos)
        rows = mod(x2.unsqueeze(1), y2.unsqueeze(1), **{k: (v.unsqueeze(1) if v.ndim == 2 else v) for k, v in pk.items()},
                   dims=[1], **call_kwargs)
        out["row_loss"] = rows.numpy()
        assert torch.equal(rows.mean(), scalar.detach()) or abs(rows.mean().item() - scalar.item()) < 1e-9
        q = mod(x2, y2, **pk, return_quantiles=True, **call_kwargs)
        uq, vq, Q, U, V = [t.numpy() for t in q]
        if store_mid:
            out.update(uq=uq, vq=vq, Q=Q, U=U, V=V)
        else:
            out.update(U_last=U[:, -1].copy(), V_last=V[:, -1].copy())
        # bit-identity of the restatement (the "port" CPU baseline)
        xp2 = mod.fixed_x if x_pos is None else x_pos
        yp2 = mod.fixed_x if y_pos is None else y_pos
        flags = dict(p=ctor.get("p", 1), square_dist=ctor.get("square_dist", False),
                     dont_normalize=ctor.get("dont_normalize", False) or call_kwargs.get("dont_normalize", False),
                     limit_quantile_range=ctor.get("limit_quantile_range", False)
                     or call_kwargs.get("limit_quantile_range", False),
                     require_sort=ctor.get("require_sort", True))
        mine = tr.sot_loss(x, y, xp2, yp2, **flags)
        assert torch.equal(mine, scalar.detach()), (name, mine.item(), scalar.item())
        mine_rows = tr.sot_loss(x2, y2, pk.get("x_pos", xp2), pk.get("y_pos", yp2), reduce=False, **flags)
        assert torch.equal(mine_rows, rows), name
    if store_inputs:
        out["x"], out["y"] = x.numpy(), y.numpy()
        if x_pos is not None:
            out["x_pos"], out["y_pos"] = x_pos.numpy(), y_pos.numpy()
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    losses, features, _ = import_reference()
    manifest = {}

    def emit(name, arrays, meta):
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        meta["scalar"] = float(arrays["scalar"])
        manifest[name] = meta
        print(f"{name:40s} scalar={meta['scalar']:.9g}")

    # 1. small fully-stored cases: mode x distribution at (4,512) -- BASELINE config 1
    pos512 = torch.linspace(0, 1, 512)
    for mode, ctor in MODES.items():
        for kind in ("uniform", "peaky", "dyadic"):
            if mode in ("p2", "p3", "p1_cut") and kind != "uniform":
                continue
            x, y = gen_inputs(kind, 4, 512, 512, 1234)
            emit(f"b4n512_{kind}_{mode}", run_case(losses, mode, x, y, pos512, pos512.clone(), ctor),
                 dict(ctor=ctor, kind=kind, seed=1234, shape=[4, 512, 512], pos="linspace"))

    # 2. edge rows (zero mass, Diracs, zero-weight runs, identical rows, sub-eps mass)
    pos64 = torch.linspace(0, 1, 64)
    for mode in ("p1", "cutoff", "nocut", "p2"):
        x, y = gen_inputs("edge", 6, 64, 64, 7)
        emit(f"edge_b6n64_{mode}", run_case(losses, mode, x, y, pos64, pos64.clone(), MODES[mode]),
             dict(ctor=MODES[mode], kind="edge", seed=7, shape=[6, 64, 64], pos="linspace"))

    # 3. n != m, odd sizes, tiny sizes
    for (n, m) in ((5, 9), (50, 70), (1, 1), (7, 7), (8, 33), (257, 257), (130, 31)):
        for mode in ("p1", "cutoff"):
            x, y = gen_inputs("uniform", 3, n, m, 100 + n)
            xp, yp = torch.linspace(0, 1, n), torch.linspace(0.1, 0.9, m)
            emit(f"nm_{n}x{m}_{mode}", run_case(losses, mode, x, y, xp, yp, MODES[mode]),
                 dict(ctor=MODES[mode], kind="uniform", seed=100 + n, shape=[3, n, m], pos="linspace/linspace(.1,.9)"))

    # 4. 3-D input [batch, time, N] + fixed_x form (metrics.py:148) + dims/hinge handled by host tests
    x, y = gen_inputs("peaky", 6, 16, 16, 5)
    x3, y3 = x.reshape(2, 3, 16), y.reshape(2, 3, 16)
    for mode in ("p1", "p2"):
        ctor = dict(MODES[mode], fixed_x=16)
        emit(f"fixedx_3d_{mode}", run_case(losses, mode, x3, y3, None, None, ctor),
             dict(ctor=ctor, kind="peaky", seed=5, shape=[2, 3, 16], pos="fixed_x"))

    # 5. per-row UNSORTED positions (require_sort=True does real work) + sort permutation
    for (n, m) in ((50, 70), (64, 64), (300, 300)):
        g = torch.Generator().manual_seed(31 + n)
        x, y = gen_inputs("uniform", 5, n, m, 31 + n)
        xp = torch.rand(5, n, generator=g)
        yp = torch.rand(5, m, generator=g)
        for mode in ("p1", "cutoff"):
            arrays = run_case(losses, mode, x, y, xp, yp, MODES[mode])
            arrays["x_sorter"] = torch.sort(xp, 1)[1].numpy()
            arrays["y_sorter"] = torch.sort(yp, 1)[1].numpy()
            emit(f"unsorted_{n}x{m}_{mode}", arrays,
                 dict(ctor=MODES[mode], kind="uniform", seed=31 + n, shape=[5, n, m], pos="rand per row"))

    # 6. the paper's real row length (n_fft 2048 -> 1025 bins; 512 -> 257), rfftfreq grid; inputs are
    #    regenerated from the seed (oracle/inputs.py; sha256 stored), outputs: row losses, grads, knife-edge CDF ends
    for N, nfft in ((1025, 2048), (257, 512)):
        pos = torch.fft.rfftfreq(nfft, 1 / 16000.0)
        pos = (pos / pos.max()).float()
        for kind in ("uniform", "peaky", "dyadic"):
            x, y = gen_inputs(kind, 16, N, N, 99)
            for mode in ("cutoff", "nocut", "p1"):
                arrays = run_case(losses, mode, x, y, pos, pos.clone(), MODES[mode], store_inputs=False,
                                  store_mid=False)
                arrays["inputs_sha256"] = np.frombuffer(bytes.fromhex(sha(x, y)), dtype=np.uint8)
                emit(f"seeded_b16n{N}_{kind}_{mode}", arrays,
                     dict(ctor=MODES[mode], kind=kind, seed=99, shape=[16, N, N], pos=f"rfftfreq({nfft})/max",
                          seeded=True))

    # 7. BASELINE config 2 rows (N=2048): 256 rows stored; plus seed-regenerated 8192-row scalars
    pos2048 = torch.linspace(0, 1, 2048)
    for kind in ("uniform", "peaky", "dyadic"):
        x, y = gen_inputs(kind, 256, 2048, 2048, 1234)
        for mode in ("p1", "cutoff", "nocut"):
            arrays = run_case(losses, mode, x, y, pos2048, pos2048.clone(), MODES[mode], store_inputs=False,
                              store_mid=False, grads=False)
            arrays["inputs_sha256"] = np.frombuffer(bytes.fromhex(sha(x, y)), dtype=np.uint8)
            emit(f"seeded_b256n2048_{kind}_{mode}", arrays,
                 dict(ctor=MODES[mode], kind=kind, seed=1234, shape=[256, 2048, 2048], pos="linspace", seeded=True))
    big = {}
    for kind in ("uniform", "peaky", "dyadic"):
        x, y = gen_inputs(kind, 8192, 2048, 2048, 1234)
        for mode in ("p1", "cutoff", "nocut"):
            with torch.no_grad():
                s = losses.Wasserstein1D(**MODES[mode])(x, y, x_pos=pos2048, y_pos=pos2048.clone())
            big[f"{kind}_{mode}"] = float(s)
            print(f"B=8192 N=2048 {kind:8s} {mode:7s} scalar={float(s):.17g}")
        big[f"{kind}_sha256"] = sha(x, y)
    manifest["_config2_b8192n2048_seed1234"] = big
    big = {}
    pos512 = torch.linspace(0, 1, 512)
    for kind in ("uniform", "peaky"):
        x, y = gen_inputs(kind, 8192, 512, 512, 1234)
        for mode in ("p1", "cutoff", "nocut"):
            with torch.no_grad():
                s = losses.Wasserstein1D(**MODES[mode])(x, y, x_pos=pos512, y_pos=pos512.clone())
            big[f"{kind}_{mode}"] = float(s)
    manifest["_config4_dense_b8192n512_seed1234"] = big

    # 8. harmonic spectra: own additive generator (f0 ~ U[40,1950] Hz, 8 partials, amps ~ U[0.4,1],
    #    4096 samples @ 16 kHz, peak 0.9: the distribution of synthetic_data.py:331-345) fed through the
    #    REFERENCE's TorchSTFT (features.py:85-113: n_fft 2048, hop 256, flattop, normalized, end-padded)
    from oracle.inputs import harmonic_audio_pair
    audio_x, audio_y = harmonic_audio_pair(nb=2, seed=11)

    tfm = features.get_transform({"type": "stft", "n_fft": 2048, "hop_length": 256, "window": "flattop"}, 16000)
    sx, sy = tfm(audio_x).contiguous(), tfm(audio_y).contiguous()
    pos = tfm.get_frequencies()
    pos = (pos / pos.max()).float()
    print("harmonic STFT spectra", tuple(sx.shape), sx.dtype)
    np.savez_compressed(os.path.join(OUT, "inputs_harmonic_stft.npz"), x=sx.numpy(), y=sy.numpy(),
                        x_pos=pos.numpy(), y_pos=pos.numpy())
    for mode in ("cutoff", "nocut", "p1"):
        emit(f"harmonic_stft_{mode}",
             run_case(losses, mode, sx, sy, pos, pos.clone(), MODES[mode], store_inputs=False, store_mid=False),
             dict(ctor=MODES[mode], kind="harmonic-stft", seed=11, shape=list(sx.shape), pos="rfftfreq/max",
                  inputs="inputs_harmonic_stft"))

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    tot = sum(os.path.getsize(os.path.join(OUT, p)) for p in os.listdir(OUT))
    print(f"{len(manifest)} entries, {tot / 1e6:.2f} MB in {OUT}")


if __name__ == "__main__":
    main()
