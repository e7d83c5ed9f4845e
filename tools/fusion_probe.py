"""Row f1 of SURVEY 8(f) for SOT-512 (n_fft 512, hop 256, 257 bins; paper-experiments/SOT-512/*/train_config.yaml:115-120): what would a
kernel that fuses the STFT-magnitude producer with the SOT training form save?  Measures, on 1024 clips x 4096 samples (16384 + 16384
frames -> 16384 rows x 257 bins), the unfused chain kernel by kernel and the two things a fusion removes:
  (a) the producer's stores of the spectra (timing-only library variant built with -DSOT_STFT_ABLATE_STORE=1: everything computed, nothing stored),
  (b) the consumer's loads of them from HBM (the same SOT launch on ONE input set, resident in L2 / Infinity Cache, against rotating sets),
plus the kernel boundary between the two (back-to-back launch of both against their separate times).
Usage on the GPU box:  python tools/fusion_probe.py [variant-library.so]     (the variant is built by tools/build_stft_variant.sh)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys; sys.path.insert(0, %(root)r)
if %(lib)r: os.environ["SOT_LIB_PATH"] = %(lib)r
import torch
from sot_amd import _native as nat, spectra
from sot_amd.losses import Wasserstein1D
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
N_FFT, HOP, CLIPS = %(n_fft)d, %(hop)d, %(clips)d
g = torch.Generator(device=dev).manual_seed(0)
tgt = [spectra.harmonic_batch(CLIPS, generator=g, device=dev) for _ in range(4)]
est = [spectra.harmonic_batch(CLIPS, generator=g, device=dev) for _ in range(4)]
win = spectra._cached_window("flattop", N_FFT, dev)
pos = spectra.unit_frequencies(N_FFT, 16000.0, dev); pos2 = pos.clone()
mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
def ev(fn, n=60):
    for i in range(20): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
specs = [nat.stft_mag_forward_pair(tgt[i], est[i], win, N_FFT, HOP) for i in range(4)]
rows = [(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])) for a, b in specs]
x2, y2, xp, yp, flags, plan, _ = mod._marshal(rows[0][0], rows[0][1], pos, pos2, {})
t_stft = ev(lambda i: nat.stft_mag_forward_pair(tgt[i %% 4], est[i %% 4], win, N_FFT, HOP))
t_sot_rot = ev(lambda i: nat.loss_and_grad(rows[i %% 4][0], rows[i %% 4][1], xp, yp, 2.0, flags, plan))
t_sot_hot = ev(lambda i: nat.loss_and_grad(rows[0][0], rows[0][1], xp, yp, 2.0, flags, plan))
def both(i):
    a, b = nat.stft_mag_forward_pair(tgt[i %% 4], est[i %% 4], win, N_FFT, HOP)
    nat.loss_and_grad(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]), xp, yp, 2.0, flags, plan)
t_both = ev(both)
gy = torch.rand_like(specs[0][1])
t_bwd = ev(lambda i: nat.stft_mag_backward(est[i %% 4], win, N_FFT, HOP, gy))
print("%(tag)s: n_fft %%d, %%d rows x %%d bins | stft pair fwd %%.1f us | sot loss+grad: rotating inputs %%.1f, one resident set %%.1f us | both back to back %%.1f us | stft bwd %%.1f us"
      %% (N_FFT, rows[0][0].shape[0], rows[0][0].shape[1], t_stft, t_sot_rot, t_sot_hot, t_both, t_bwd))
'''
variant = sys.argv[1] if len(sys.argv) > 1 else ""
for n_fft, hop, clips in ((512, 256, 1024), (2048, 256, 256)):
    for tag, lib in (("product ", ""), ("no-store", variant)):
        if tag == "no-store" and not variant:
            continue
        for rep in range(2):
            r = subprocess.run([sys.executable, "-c", CODE % dict(root=ROOT, lib=lib, n_fft=n_fft, hop=hop, clips=clips, tag=tag)], capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
