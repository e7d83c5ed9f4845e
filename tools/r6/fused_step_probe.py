"""The paper's loss step as ONE autograd node (spectra._fused_mix_step / csrc/sot_torch_glue.cpp: MixLossStep) against the module-by-module
composition: loss and gradient differences, then GPU time replayed from a HIP graph and host-launched time, at 64 and 256 clips.
python3 tools/r6/fused_step_probe.py [modes: m (module by module), n (one node); default mn] [compare=1]"""
import faulthandler
import functools
import os
import sys
import time

faulthandler.enable()
print = functools.partial(print, flush=True)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from sot_amd import spectra
from sot_amd.losses import MixOfLosses, MSSLoss, Wasserstein1D

dev = torch.device("cuda:0")
mss = MSSLoss(fft_sizes=(2048, 1024, 512, 256, 128, 64), loss_type="L1", mag_weight=1, logmag_weight=0).to(dev)
sot = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True, require_sort=True).to(dev)
mix = MixOfLosses([mss, sot], [0.05, 1]).to(dev)
freqs = torch.fft.rfftfreq(2048, d=1.0 / 16000.0).to(dev)
seed = torch.ones((), device=dev)


def timed(fn, n=200, warm=30):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def replayed(step_fn):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step_fn(0)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step_fn(0)
    return timed(lambda i: graph.replay())


MODES = {"m": ("module by module", False), "n": ("one node", True)}
modes = [MODES[c] for c in (sys.argv[1] if len(sys.argv) > 1 else "mn")]
compare = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
for clips in (64, 256):
    gen = torch.Generator(device=dev).manual_seed(1000 + clips)
    x = spectra.harmonic_batch(clips, generator=gen, device=dev)
    hats = [spectra.harmonic_batch(clips, generator=gen, device=dev).requires_grad_(True) for _ in range(2)]

    def step(i, fused):
        e = hats[i % 2]
        e.grad = None
        loss = spectra.trainer_loss_step(mix, x, e, positions=freqs, fused=fused)
        loss.backward(seed)
        return loss.detach()

    res = {}
    for name, fused in (MODES.values() if compare else ()):
        loss = step(0, fused)
        res[name] = (float(loss.detach()), hats[0].grad.clone())
        del loss   # (a loss kept alive keeps the estimate's AccumulateGrad node of the DEFAULT stream alive: a capture on another stream then dies in capture_end)
    base_l, base_g = res.get("module by module", (0, 0))
    for name, (l, g) in res.items():
        print(f"{clips} clips, {name:16s}: loss {l:.9g} (rel diff {abs(l - base_l) / abs(base_l):.2e}), gradient max diff / peak "
              f"{float((g - base_g).abs().max() / base_g.abs().max()):.2e}")
    for name, fused in modes:
        eager = timed(lambda i: step(i, fused))
        graph = replayed(lambda i: step(i, fused))
        print(f"{clips} clips, {name:16s}: eager {eager:7.1f} us, graph replay {graph:7.1f} us")
