"""Interleaved A/B of library variants (tools/ablate_libs/<name>.so, full builds: `python -c "import sot_amd; sot_amd.build.build(
extra_flags=[...], out=...)"`) on the STFT kernels: forward, forward of a pair, backward; 256 clips x 4096 samples, n_fft 2048 / hop 256
(config 5) and n_fft 512 / hop 128."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:]
for rnd in range(2):
    for name in names:
        code = f"""
import os, sys; sys.path.insert(0, {ROOT!r})
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import torch
from sot_amd import _native as nat, spectra
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
a = torch.rand(256, 4096, device=dev, generator=g) - 0.5; b = torch.rand(256, 4096, device=dev, generator=g) - 0.5
def ev(fn, n=100):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for n_fft, hop in ((2048, 256), (512, 128)):
    win = spectra._cached_window('flattop', n_fft, dev)
    frames = -(-4096 // hop)
    gm = torch.rand(256, frames, n_fft // 2 + 1, device=dev, generator=g)
    out.append((n_fft, round(ev(lambda: nat.stft_mag_forward(a, win, n_fft, hop)), 1), round(ev(lambda: nat.stft_mag_forward_pair(a, b, win, n_fft, hop)), 1),
                round(ev(lambda: nat.stft_mag_backward(a, win, n_fft, hop, gm)), 1)))
print(' | '.join(f'n_fft {{n}}: fwd {{f}} pair {{p}} bwd {{bw}} us' for n, f, p, bw in out))
"""
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print(f"{name:10s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
